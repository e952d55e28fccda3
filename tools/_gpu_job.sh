mkdir -p gpurun_out/r2x
for s in 0 30 60 100 0 60; do
  echo "stagger $s"; VDK_GEMM_STAGGER=$s python bench.py --no-cbir --no-cpu-baseline --no-parity --steps 20 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline'].get('by_variant', ''))
"
done 2>&1 | tee gpurun_out/r2x/stagger.txt
