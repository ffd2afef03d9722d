cd /root/repo
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
timeout 300 python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
