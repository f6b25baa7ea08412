timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-energy-check 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n13', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['lds_bytes_per_walker'])"
done
timeout -k 10 200 python bench.py --n 29 --batch 2048 --Emax 25 --no-cpu-baseline --no-energy-check 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n29', d['value'], d['ms_per_step'], d['roofline']['frac'])"
timeout -k 10 200 python bench.py --n 57 --batch 512 --Emax 49 --no-cpu-baseline --no-energy-check 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n57', d['value'], d['ms_per_step'], d['roofline']['frac'])"
