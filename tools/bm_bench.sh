cd /root/repo
for c in 1 2 3 4; do
echo chunks=$c bm=1; SMPLFIT_BM=1 SMPLFIT_CHUNKS=$c python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-140
done
echo chunks=2 bm=0; SMPLFIT_CHUNKS=2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-140
