cd /root/repo
timeout 600 python -m pytest tests/test_gpu_trainer.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | cut -c100-180; done
cp socialways_amd/trainer.py /tmp/t_new.py; cp tools/_t_old.py socialways_amd/trainer.py; for i in 1 2; do timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | cut -c100-180; done; cp /tmp/t_new.py socialways_amd/trainer.py
