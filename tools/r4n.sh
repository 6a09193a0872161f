for v in 3 1 2 0; do echo NOPV=$v; BODAHIP_EXTRA_DEFS="-DNOPV=$v" python tools/k1q_debug.py 2>&1 | grep "^bad"; BODAHIP_EXTRA_DEFS="-DNOPV=$v" CASE=9:96:55:55:96 python tools/k1q_debug.py 2>&1 | grep "^bad"; done
python -m pytest tests/test_gpu_parity.py -x -q -k "k1_stream" 2>&1 | tail -3
BODAHIP_NO_K1_QUAD=1 SHAPES="256:96:55:96,128:96:55:96" SPECS="off,q4x3x8,q4x3x4,q4x3x12,q4x3x16,q8x3x8,q2x3x8,q4x3x8x1,q8x3x8x1,q8x3x16x1" python tools/k1s_probe.py 2>&1 | tee gpurun_out/s2/k1q_probe.txt
