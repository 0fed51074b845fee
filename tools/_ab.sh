export TMPDIR=/tmp
python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "spade or spectral or sparse" 2>&1 | tail -2
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > /tmp/b.json 2>/dev/null
grep -h "class_table\|sn_gemv\|uniform" /tmp/kt/*/*kernel_stats.csv /tmp/kt/*kernel_stats.csv 2>/dev/null | cut -c1-170
python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
