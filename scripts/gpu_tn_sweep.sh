cd /root/repo
for t in 128 192 256 320 384 512; do echo "== target $t"; MAED_TN_TARGET_WGS=$t WGRAD_STE=1 timeout 200 python scripts/wgrad_micro.py 10 2>&1 | grep -E "STE wgrad|H= 14|H= 28 I= 128|H= 56 I= 256 O=  64|backbone total" | cut -c1-100; done
