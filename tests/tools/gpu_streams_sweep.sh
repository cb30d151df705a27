# Run ON the GPU box: the headline step with other stream / context counts (LM_STREAMS parts per context, --inflight contexts)
for s in 2 3 4; do for f in 2 3; do
  echo "== LM_STREAMS=$s inflight=$f"
  LM_STREAMS=$s python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-end-to-end --inflight $f 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['config'].get('streams_per_context'))"
done; done
