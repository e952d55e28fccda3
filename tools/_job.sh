cd /root/repo
timeout 300 python tools/bench_swin.py 128 8 native 2>/dev/null
timeout 900 python bench.py --no-cbir --no-cfg5 --no-swin --no-cpu-baseline > gpurun_out/bench_mid.json 2> gpurun_out/bench_mid.err; python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/bench_mid.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','dtype')}, d.get('other_operand',{}).get('ms_per_step'), d.get('parity',{}).get('tolerance_met'))
PY
