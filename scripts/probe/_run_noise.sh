cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 250 python tests/diagnostics/scalar_grad_noise.py 2>&1 | grep -v Initializing | tail -12 || exit 1
LG_ROWS_XJ1=1 timeout -k 10 250 python tests/diagnostics/scalar_grad_noise.py 2>&1 | grep -v Initializing | tail -12 || exit 1
LG_NO_ZN=1 timeout -k 10 250 python tests/diagnostics/scalar_grad_noise.py 2>&1 | grep -v Initializing | tail -12 || exit 1
LG_N3W_TH8=1 LG_NO_UP4_PAIR=1 timeout -k 10 250 python tests/diagnostics/scalar_grad_noise.py 2>&1 | grep -v Initializing | tail -12 || exit 1
