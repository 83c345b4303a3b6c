R=${GRAFT_REPO_ROOT:-/root/repo}; P="python3 $R/profiles/r04/malloc_flags_probe.py"
echo "== $(date +%H:%M:%S)"
for a in product own 0 1 3 4; do $P $a 2>&1 | tail -n 1; done
$P product 1 2>&1 | tail -n 1; $P product 3 2>&1 | tail -n 1; $P 3 3 2>&1 | tail -n 1; $P product 2>&1 | tail -n 1
