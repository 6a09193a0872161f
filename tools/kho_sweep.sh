#!/bin/bash
# Sequential K hand-off (round 5, gemm_conv_f32.hip -DKHO=1): parity tests, then the tile-starved / unevenly dealt fp32 layers of NiN and AlexNet under
# tile x K segments (eleventh tile field), against the planner's choice.   gpurun --timeout 1800 -- 'bash tools/kho_sweep.sh'
O=gpurun_out/kho; mkdir -p $O
timeout 600 python -m pytest "tests/test_gpu_parity.py" -q -m gpu -x -k "hand_off" > $O/tests.log 2>&1; echo "tests exit $?" >> $O/tests.log; tail -4 $O/tests.log
gen() {  # base tiles (ten fields) x segment counts
  out=""; for t in $1; do for s in $2; do out="$out,${t}x$s"; done; done; echo "${out#,}"
}
P3="128x128x36x2x2x2x1x32x1x0 64x256x36x1x4x2x1x32x1x0 128x256x36x2x4x1x1x32x1x0 64x128x36x1x4x2x1x32x1x0"
K1="128x128x16x2x2x2x1x32x1x0 128x128x32x2x2x2x1x32x1x0 128x256x16x2x4x1x1x32x1x0 64x128x16x1x4x2x1x32x1x0 64x256x16x1x4x2x1x32x1x0"
for b in 128 256; do
  timeout 900 python tools/tile_sweep.py --workload nin --batch $b --ops 9 --tiles "$(gen "$P3" "2 3 4 6 8")" > $O/nin${b}_conv4.txt 2>&1
  timeout 900 python tools/tile_sweep.py --workload nin --batch $b --ops 10,11 --tiles "$(gen "$K1" "2 4 8")" > $O/nin${b}_cccp78.txt 2>&1
  timeout 900 python tools/tile_sweep.py --workload nin --batch $b --ops 7 --tiles "$(gen "$K1" "2 3 4")" > $O/nin${b}_cccp5.txt 2>&1
  timeout 900 python tools/tile_sweep.py --workload nin --batch $b --ops 4 --tiles "$(gen "$K1" "2 4")" > $O/nin${b}_cccp3.txt 2>&1
  timeout 900 python tools/tile_sweep.py --workload nin --batch $b --ops 6 --tiles "$(gen "$P3" "2 4")" > $O/nin${b}_conv3.txt 2>&1
done
timeout 900 python tools/tile_sweep.py --workload alexnet --batch 256 --ops 2,3,4 --tiles "$(gen "$P3" "2 3 4 6")" > $O/alex256_conv345.txt 2>&1
grep -h "^op" $O/*.txt | sort -k1,2 -s | awk '{print}' | head -400
