#!/bin/bash
# config 3 with the silhouette gradients through the gather kernel (BF_MASK_FOLD=gather) and through the fixed-point sums, interleaved
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/${1:-fold}
for rep in 1 2; do for v in gather sums; do
  BF_MASK_FOLD=$v timeout 300 python bench.py --config 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['ms_per_step_parts_rank0']; print('BF_MASK_FOLD=$v cfg3', '%.2f fps' % d['value'], 'fit %.3f ms' % p['fit_ms'], 'per it %.4f' % p['ms_per_fit_iteration'])"
done; done | tee gpurun_out/${1:-fold}/ab.txt
python -m pytest tests -q -m gpu -x -k "mask or smplx or configs_full or contour" 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
