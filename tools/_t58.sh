timeout 1500 python -m pytest tests/test_gpu_hmm.py -x -q -m gpu 2>&1 | tail -2
for rows in 2000000 200000 50000; do
timeout 600 python tools/bench_hmm.py --classes 128 --degree 8 --rows $rows --no-cpu --no-viterbi --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('K128 rows $rows', round(d['ms_per_step'],2), d.get('boundary_pass'))"
done
