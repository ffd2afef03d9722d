cd /root/repo
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])"; done
SW_FORCE_DIST=1 timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dist1', d['value'], d['roofline']['avg_launch_ms'])"
