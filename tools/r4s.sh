mkdir -p gpurun_out/s10
BODAHIP_NO_K1_QUAD=1 SHAPES="256:256:27:256,256:384:13:384,128:256:27:256" SPECS="off,q4x4x8x1,q4x4x4x1,q8x4x8x1,q4x3x8x1,q4x2x8x2,q4x2x8x1,q4x4x16x1" python tools/k1s_probe.py 2>&1 | tee gpurun_out/s10/k1q_big.txt
