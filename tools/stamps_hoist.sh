# diagnostic: the layer-0 edge launch with every item run twice through the same code (-DPF_STAMPS -DPF_TWICE build):
# first pass (items from 0) against second pass (items from 100000) -- instruction-cache check
export PFDYN_LIB=$PWD/pharmacophore-diffusion_amd/csrc/variants/libpfdyn_twice.so
export PFDYN_NO_L0_HOIST=1 PFDYN_RG_SPLIT_MAX=0
for b in 8 32; do
echo "#### B=$b rows 4"; B=$b PFDYN_RG2_ROWS_MIN=1000000000 STAMP_OFFSETS=0,100000 python tools/stamps_rg.py 2>&1 | grep -A4 "launch 0"
echo "#### B=$b rows 8"; B=$b PFDYN_RG2_ROWS_MIN=0 STAMP_OFFSETS=0,100000 python tools/stamps_rg.py 2>&1 | grep -A4 "launch 0"
done
