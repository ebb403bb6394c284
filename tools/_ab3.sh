set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python -m pytest tests/test_gpu_gemm_staged.py -q 2>&1 | tail -5
timeout -k 10 120 python tools/staged_time.py 100000 1024 512
timeout -k 10 120 python tools/staged_time.py 100000 512 128
timeout -k 10 120 python tools/staged_time.py 1000000 1024 512
DETAIL=1 python tools/phase_ab.py cfg4 10 2>&1 | grep -v amdgpu.ids
