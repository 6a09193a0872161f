# GPU tool: layers the planner gives the 256 x 128 tile (64 x 128 wave tiles), against the 128 x 128 tile; nets
export TILES="auto 128x128x0x4x1x2"
BATCH=256 SEL=1 python tools/nhwc_sweep.py alexnet 8 2>&1 | tail -1
BATCH=128 SEL=3 python tools/nhwc_sweep.py nin 8 2>&1 | tail -1
BATCH=64 SEL=3 python tools/nhwc_sweep.py nin 8 2>&1 | tail -1
python -m pytest tests/test_gpu_nhwc.py -x -q -k "patch_kernel_tiles and direct" 2>&1 | tail -1
for w in nin-net alexnet-net; do
python bench.py --workload $w --dtype bf16 --layout nhwc --no-cpu-baseline --steps 20 --warmup 5 --graph --parallel-branches 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w',d['value'],d['images_per_s'],d['ms_per_step'],r['conv_ms'],r['non_conv_ms'],r['frac'])"
done
