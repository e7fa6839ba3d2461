# diagnostic: stamps of the layer-0 edge launch, full-chain items and hoisted items, per launch policy
export PFDYN_LIB=$PWD/pharmacophore-diffusion_amd/csrc/variants/libpfdyn_stamps.so
echo "#### no hoist"; PFDYN_NO_L0_HOIST=1 STAMP_OFFSETS=0,500 python tools/stamps_rg.py 2>&1 | grep -A7 "launch 0"
echo "#### hoist rows 8/8"; STAMP_OFFSETS=0,500 python tools/stamps_rg.py 2>&1 | grep -A7 "launch 0"
echo "#### hoist rows 4/4"; PFDYN_L0_RGA=1 PFDYN_L0_RGP=1 STAMP_OFFSETS=0,1000,1500 python tools/stamps_rg.py 2>&1 | grep -A7 "launch 0"
echo "#### hoist rows 4/8"; PFDYN_L0_RGA=1 PFDYN_L0_RGP=2 STAMP_OFFSETS=0,800,1100 python tools/stamps_rg.py 2>&1 | grep -A7 "launch 0"
