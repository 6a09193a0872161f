# GPU tool: patch-fragment read-ahead depth x tiles (incl. one-wave-per-SIMD wave tiles) at 256 and 64 images
export TILES="auto 128x128x0x4x1x2 256x128x0x4x1x1 128x256x0x4x1x1 256x256x0x4x2x1 128x128x0x4x1x1"
for bpf in 1 2 3; do
  echo "== BPF=$bpf"
  export BODAHIP_EXTRA_DEFS="-DBPF=$bpf" BODAHIP_CACHE_DIR=/tmp/kc_bpf$bpf
  BATCH=256 SEL=1,2,3 python tools/nhwc_sweep.py alexnet 8 2>&1 | tail -3
  SEL=7,12 python tools/nhwc_sweep.py resnet-50 12 2>&1 | tail -2
done
