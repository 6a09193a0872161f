# GPU tool: AlexNet fc6 / fc7 / fc8 at 256 images on the channels-last bf16 implicit GEMM: tiles x K slices x ring depth
export TILES="auto 128x128x64x2x2x2x4x32x2 128x128x64x2x2x2x8x32x2 128x128x64x2x2x2x16x32x2 64x128x64x1x4x2x8x32x3 64x128x64x1x4x2x16x32x3 64x64x64x2x2x2x8x32x3 64x64x64x2x2x2x16x32x3 64x64x64x2x2x2x32x32x3 128x128x32x2x2x2x8x32x4 128x128x32x2x2x2x16x32x4 64x256x64x1x4x2x8x32x2 64x256x64x1x4x2x16x32x2 32x256x64x1x4x2x16x32x3"
BATCH=256 SEL=5,6,7 python tools/nhwc_sweep.py alexnet 12 2>&1 | tail -4
