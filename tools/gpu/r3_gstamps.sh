cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
IONODE_LIB=$GRAFT_REPO_ROOT/neural-ode-ion-channels_amd/variants/gstamps/libionode.so timeout -k 10 300 python3 tools/bench_grad.py --reps 1 2>&1 | grep -E "WALK STAMPS|backward_s" | tail -8 | cut -c1-300
