cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2 3; do python scripts/perf_sampler.py 2>&1 | tail -1; done
