cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
timeout 600 python tools/dbg_conv3p.py 2>&1 | tail -60
