cd /root/repo
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python tools/bench_gemm_swin.py gpurun_out/gemm_swin_tn2.json 3 tn 401408:128:2,100352:256:2 2>&1 | grep -v amdgpu.ids
