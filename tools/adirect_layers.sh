# GPU tool: every patch-kernel layer of the two config-5 lists, staged vs direct, planner tiles
for ad in 0 1; do export BODAHIP_NHWC_ADIRECT=$ad BODAHIP_CACHE_DIR=/tmp/kc_adab$ad
echo "== direct=$ad"
TILES=auto python tools/nhwc_sweep.py resnet-50 12 2>&1 | grep -v "k1s" 
TILES=auto python tools/nhwc_sweep.py googlenet_conv 12 2>&1 | grep -v "k1s"
done
