#!/bin/bash
# Input-patch kernel (conv_nhwc_patch_bf16.hip, direct filter path): 64-row WAVE tiles (half the LDS bytes per MFMA) in SMALL workgroups (one or two waves), against the
# planner's choice, on the 3x3 / 5x5 layers of the config-5 nets at 64 images.   gpurun --timeout 900 -- 'bash tools/patch_wave_sweep.sh'
O=gpurun_out/pwave; mkdir -p $O
export TILES="auto 128x128x0x2x1x2 64x256x0x1x2x2 64x128x0x1x1x2 64x128x0x1x1x4 64x128x0x1x2x2 128x64x0x2x1x2 128x128x0x2x2x2 128x256x0x2x2x2 256x128x0x4x1x2 128x128x0x4x1x2"
SEL=3,7,12,17 python tools/nhwc_sweep.py resnet-50 14 > $O/resnet.txt 2>&1; grep -v "^\[" $O/resnet.txt | tail -6
SEL=2,6,7,11,12,17,30,39,45,51 python tools/nhwc_sweep.py googlenet_conv 14 > $O/googlenet.txt 2>&1; grep -v "^\[" $O/googlenet.txt | tail -11
