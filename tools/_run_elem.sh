cd $GRAFT_REPO_ROOT
for c in 0 3 0 3; do ISO_CFG=$c ISO_EPI=1 timeout 300 python tools/gemm_iso.py 2>&1 | grep -v amdgpu.ids; done
