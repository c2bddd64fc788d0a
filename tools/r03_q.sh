python -m pytest tests/test_cpp_frontend.py tests/test_gpu_parity.py -m gpu -x -q -k "frontend or cpp or context or pipeline or golden or config2" 2>&1 | tail -2
echo "== latency schedule"; python tools/dev_sync_latency.py merkle29 mimc11 14 16 18 20 2>&1 | grep sync
echo "== overlap schedule"; ZK_NO_LATENCY_SCHED=1 python tools/dev_sync_latency.py 16 18 20 2>&1 | grep sync
