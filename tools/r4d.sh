#!/bin/bash
# store cache-policy sweep on NiN cccp1 (55x55 planes, 96 -> 96 @256) and cccp3 (27x27), then the new tests
mkdir -p gpurun_out/r4d; O=gpurun_out/r4d
for aux in 0 1 2 3 16 17 18 19; do
  echo "== ST_AUX=$aux" >> $O/aux.txt
  BODAHIP_EXTRA_DEFS="-DST_AUX=$aux" SHAPES="256:96:55:96,256:256:27:256" SPECS="off" SETTLE=200 timeout 120 python tools/k1s_probe.py >> $O/aux.txt 2>&1
done
cat $O/aux.txt
python -m pytest tests/test_gpu_nhwc.py tests/test_gpu_zz_properties.py tests/test_gpu_zz_bench.py -m gpu -q --durations=12 -p no:cacheprovider -x -k "edge_free or timing or eight_ranks or wide or 520 or 2100 or default_line" > $O/pytest.log 2>&1; echo rc=$? >> $O/pytest.log; tail -25 $O/pytest.log
