cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 300 python bench.py --steps 300 --warmup 30 2>&1 | tail -1
timeout 300 python bench.py --workload c4 --steps 20 --warmup 5 2>&1 | tail -1
