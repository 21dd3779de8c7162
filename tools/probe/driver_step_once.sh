python - <<'PY'
import torch, bench, json
from geometrics_amd import gemm_tuning
gemm_tuning.enable()
print(json.dumps(bench.driver_step_times(torch.device("cuda:0")), indent=1))
PY
