# Runs ON THE GPU BOX: what the FIRST build of a process pays (the executable's case).  bench.py's warm-up step is such a build:
# its stage times against the later steps' (CORA_WARM_SYNC=1: the host buffers are prepared before the kernels instead of beside them).
set -e
cd $GRAFT_REPO_ROOT
for v in "" ${FIRST_BUILD_VARIANTS}; do
 echo "== $v"
 env $v CORA_TRACE=1 timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --no-extras 2>&1 >/dev/null | grep "stages:\|\] run \|warm:\|items: waited" | cut -c1-200
done
