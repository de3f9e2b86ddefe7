#!/bin/bash
# A/B of the uniform-ring depth of the (2,8) prefill GEMM tile against the shipped library.  The experiment
# libraries are built first (not tracked), e.g. for depth 4:
#   cd kuiperllama_amd/lib && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off \
#     -DKH_PG_RING_D16=4 -c ../csrc/kh_model_prefill.hip -o /tmp/pf.o && \
#   hipcc --offload-arch=gfx950 -shared -fPIC -o exp_ring4.so kh_ops.o kh_model_load.o kh_model_step.o /tmp/pf.o \
#     kh_model_profile.o kh_tokenizer.o kh_bpe.o        (KH_PG_RING_D16=0 = the phase scheme)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r3_prefill_ring28b.txt
: > $O
export KH_PT_SIZES=256,512
L=$PWD/kuiperllama_amd/lib
M="llama3.2-1b tinyllama-1.1b qwen2.5-0.5b llama2-7b"
timeout 600 python tools/prefill_time.py phases $M 2>/dev/null >> $O
KH_LIB=$L/exp_ring2.so timeout 600 python tools/prefill_time.py "ring D=2" $M 2>/dev/null >> $O
timeout 600 python tools/prefill_time.py phases-again $M 2>/dev/null >> $O
KH_LIB=$L/exp_ring2.so timeout 600 python tools/prefill_time.py "ring D=2 again" $M 2>/dev/null >> $O
cat $O
