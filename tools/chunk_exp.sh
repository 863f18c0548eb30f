for c in 0 64 32 16; do
  echo "== chunk $c"; timeout -k 10 200 python bench.py --steps 5 --warmup 2 --chunk $c --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], {k:(v['avg_ms'], v['tflops']) for k,v in d['kernels'].items()})"
done
