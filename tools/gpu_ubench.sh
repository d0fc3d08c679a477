#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 120 build/valu_ubench 2>&1 | tee gpurun_out/valu_ubench.log
