mkdir -p gpurun_out/s12
(python tools/fuzz_conv.py 400 11; echo "rc=$?") 2>&1 | tail -4
(python tools/fuzz_conv.py 120 12 big; echo "rc=$?") 2>&1 | tail -4
(BODAHIP_FC=32x64x32x2 python tools/fuzz_conv.py 300 13; echo "rc=$?") 2>&1 | tail -3
(BODAHIP_FC=64x32x64x4 python tools/fuzz_conv.py 100 14 big; echo "rc=$?") 2>&1 | tail -3
(BODAHIP_RDEC=96x256x1x4x2 python tools/fuzz_conv.py 300 15; echo "rc=$?") 2>&1 | tail -3
