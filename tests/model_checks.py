"""Model-level comparisons shared by the GPU tests: how far logits may be from their reference, and what
"the same arg-max" means for bf16 logits.

Every comparison is made through `check_logits(name, got, want)`.  The bound it asserts is twice the largest
max|got - want| / max|want| that has been OBSERVED for that comparison on an MI355X (`MEASURED`, copied from
profiles/r2_model_tolerances.json, which a run of the GPU tests regenerates as
gpurun_out/model_tolerances_observed.json); a comparison with no measurement yet falls back to `UNMEASURED`
and says so in the record.  Both sides hold bf16 weights and accumulate in fp32; what differs is the order of
the sums and where intermediate results are rounded to bf16, hence errors of a few bf16 ulps of the largest
logit, not of 1e-3 of each logit (the reference's own bf16 path has the same spread against an fp32 run).
"""
from __future__ import annotations

import json
import os

import numpy as np

UNMEASURED = 2.0 ** -6
FLOOR = 2.0 ** -8     # one bf16 ulp of the largest logit, at best: the bound where the observed difference was 0
FLOOR_F16 = 2.0 ** -11   # the same for the comparisons that run in float16 (names starting "tiny_f16")

# name -> observed max|diff| / max|want| on MI355X
MEASURED: dict = {
    "config1_cpu_fp32_engine_vs_hip_engine": 0.0188121,
    "config3_prefix_reuse_vs_recompute": 0,
    "config5_chunked_vs_one_step_prefill": 0,
    "forced_collectives_vs_one_rank": 0.00877193,
    "fused_vs_library[0.6B/tp2]": 0.00506073,
    "fused_vs_library[0.6B/tp4]": 0.00408021,
    "fused_vs_library[0.6B/tp8]": 0,
    "fused_vs_library[config3_qwen3_8b_tp1]": 0.00865801,
    "fused_vs_library[config4_qwen3_8b_tp8]": 0.00775194,
    "fused_vs_library[config5_qwen3_32b_tp4_stage]": 0.00843882,
    "fused_vs_library[hidden512]": 0.00441505,
    "g7_k_cache_vs_reference": 0.00680272,
    "g7_reference_vs_hip_eager": 0.00840336,
    "g7_reference_vs_hip_unfused": 0.00840336,
    "g7_unfused_vs_fused": 0,
    "pp2_vs_one_rank": 0,
    "tiny_f16_hip_vs_f16_cpu": 0.00158144,
    "tiny_f16_hip_vs_fp32_cpu": 0.00143307,
    "tp2_vs_one_rank": 0.011239,
    "tp4_vs_one_rank": 0.0111607,
}

_OBSERVED: dict = {}
_OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "model_tolerances_observed.json")


def record(name: str, value: float):
    """Keep the largest `value` seen under `name` in gpurun_out/model_tolerances_observed.json."""
    _OBSERVED[name] = max(_OBSERVED.get(name, 0.0), float(value))
    try:
        os.makedirs(os.path.dirname(_OUT), exist_ok=True)
        old = {}
        if os.path.exists(_OUT):
            with open(_OUT) as f:
                old = json.load(f)
        old.update(_OBSERVED)
        with open(_OUT, "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)
    except OSError:
        pass


def bound(name: str) -> float:
    if os.environ.get("NVX_MEASURE_TOLERANCES") == "1":    # a measuring run: record, assert nothing about the size
        return float("inf") if name not in MEASURED else max(1.0, 2.0 * MEASURED[name])
    floor = FLOOR_F16 if name.startswith("tiny_f16") else FLOOR
    return max(2.0 * MEASURED[name], floor) if name in MEASURED else UNMEASURED


def bf16_ulp(x: float) -> float:
    """Spacing of bf16 numbers around |x| (8 significant bits)."""
    x = abs(float(x))
    return 2.0 ** (np.floor(np.log2(x)) - 7) if x > 0 else 2.0 ** -133


def argmax_agrees(got: np.ndarray, want: np.ndarray, slack: float) -> bool:
    """Row-wise arg-max agreement, except where the two candidates are tied in `want` to within `slack`
    (absolute): a tie that rounding may break either way."""
    ga, wa = got.argmax(-1), want.argmax(-1)
    for r in np.nonzero(ga != wa)[0]:
        if want[r, wa[r]] - want[r, ga[r]] > slack:
            return False
    return True


def check_close(name: str, got, want, what=""):
    """max|got - want| / max|want| against twice its observed value (no arg-max part)."""
    got, want = np.asarray(got, dtype=np.float32), np.asarray(want, dtype=np.float32)
    err = float(np.abs(got - want).max()) / float(np.abs(want).max())
    record(name, err)
    assert err <= bound(name), f"{name} {what}: max|diff|/max|ref| = {err:.3e} > {bound(name):.3e}"
    return err


def check_logits(name: str, got: np.ndarray, want: np.ndarray, what="", tie_ulps: float = 1.0):
    got, want = np.asarray(got, dtype=np.float32), np.asarray(want, dtype=np.float32)
    scale = float(np.abs(want).max())
    err = float(np.abs(got - want).max()) / scale
    record(name, err)
    lim = bound(name)
    assert err <= lim, f"{name} {what}: max|diff|/max|logit| = {err:.3e} > {lim:.3e}"
    # arg-max must agree on every row, except where the reference's own top two are within one bf16 ulp (of the
    # largest logit) of each other: the logits are bf16 numbers, so that is a tie the output rounding decides
    assert argmax_agrees(got.reshape(-1, got.shape[-1]), want.reshape(-1, want.shape[-1]), tie_ulps * bf16_ulp(scale)), \
        f"{name} {what}: arg-max differs on a row whose top two logits are more than {tie_ulps} bf16 ulp apart"
    return err
