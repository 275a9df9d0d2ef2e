cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
PY
for g in 4096 2048 1024 768 512; do echo "grid=$g"; LG_PATCH_GRID=$g timeout -k 10 100 python scripts/bench_patch.py; done
