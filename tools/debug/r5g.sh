for T in 128 257 513; do for B in 1 4 8 15; do
RDPN6D_H2_SPLIT_TILES=$T python bench.py --no-cpu-baseline --batch $B --steps 300 --preheat 0.5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('split_tiles=$T B=$B', d['ms_per_step'])"
done; done
