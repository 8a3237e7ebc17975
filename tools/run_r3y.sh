# what the per-launch HIP events of the dominant family cost the timed region: alternating runs with and without them
for r in 1 2 3; do
  for cfg in 1 2; do
    for ev in on off; do
      if [ $ev = off ]; then export ENDO_BENCH_NO_EVENTS=1; else unset ENDO_BENCH_NO_EVENTS; fi
      python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $cfg events $ev  %.3f ms/step  %.1f frame-pairs/s' % (d['ms_per_step'], d['value']))"
    done
  done
done
