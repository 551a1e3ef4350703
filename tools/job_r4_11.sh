mkdir -p gpurun_out
for cfg in 0 6; do echo "== tile_cfg $cfg"; ISO_CFG=$cfg LAT_B=1,8 python tools/latency_bench.py 24 2>&1 | tail -1; done | tee gpurun_out/lat_halo.txt
