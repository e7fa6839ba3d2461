# diagnostic: per-block cycles of the layer-0 edge launch against the number of busy waves (contention check)
export PFDYN_LIB=$PWD/pharmacophore-diffusion_amd/csrc/variants/libpfdyn_stamps.so
export PFDYN_NO_L0_HOIST=1 PFDYN_RG_SPLIT_MAX=0
for b in 2 8 16 32 64; do
echo "#### B=$b rows 8"; B=$b PFDYN_RG2_ROWS_MIN=0 python tools/stamps_rg.py 2>&1 | grep -A4 "launch 0"
echo "#### B=$b rows 4"; B=$b PFDYN_RG2_ROWS_MIN=1000000000 python tools/stamps_rg.py 2>&1 | grep -A4 "launch 0"
done
