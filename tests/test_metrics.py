"""Validation metrics (reference loss_functions.py:355-467): the oracle restatement and the product
functions against fixtures written by the unmodified reference (tests/golden/metrics.npz, oracle/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import metrics as OM
from oracle.make_golden import metric_inputs


@pytest.fixture(scope="module")
def gold(golden_dir):
    import os
    return np.load(os.path.join(golden_dir, "metrics.npz"))


def _check(fns, m, gold, tol):
    def close(a, b):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        assert np.max(np.abs(a - b)) <= tol * max(1.0, np.max(np.abs(b))), (a, b)
    close(fns["flow_diff"](m["gt"], m["rigid"]).detach().cpu().numpy(), gold["flow_diff"])
    close(fns["compute_epe"](m["gt"], m["rigid"]), gold["epe3"])
    close(fns["compute_epe"](m["gt"][:, :2].contiguous(), m["nonrigid"]), gold["epe2"])
    close(fns["outlier_err"](m["gt"], m["rigid"]), gold["outlier"])
    close(fns["compute_all_epes"](m["gt"], m["rigid"], m["nonrigid"], m["mask"]), gold["all_epes"])
    close([float(v) for v in fns["compute_errors"](m["dgt"], m["dpred"])], gold["errors_crop"])
    close([float(v) for v in fns["compute_errors"](m["dgt"], m["dpred"], crop=False)], gold["errors_nocrop"])


def test_oracle_metrics_match_reference(gold):
    fns = {k: getattr(OM, k) for k in ("flow_diff", "compute_epe", "outlier_err", "compute_all_epes", "compute_errors")}
    _check(fns, metric_inputs(), gold, 1e-6)


def test_product_metrics_match_reference_cpu(gold):
    from cc_amd import loss_functions as LF
    fns = {k: getattr(LF, k) for k in ("flow_diff", "compute_epe", "outlier_err", "compute_all_epes", "compute_errors")}
    _check(fns, metric_inputs(), gold, 1e-6)


def test_product_metrics_async_flavour(gold):
    from cc_amd import loss_functions as LF
    m = metric_inputs()
    out = LF.compute_all_epes(m["gt"], m["rigid"], m["nonrigid"], m["mask"], sync=False)
    assert all(torch.is_tensor(v) and v.dim() == 0 for v in out)
    assert np.allclose([float(v) for v in out], gold["all_epes"], rtol=1e-6)


def test_edge_aware_smoothness_per_pixel():
    from cc_amd import loss_functions as LF
    g = torch.Generator().manual_seed(0)
    img, pred = torch.rand(2, 3, 9, 9, generator=g), torch.rand(2, 1, 9, 9, generator=g)
    with pytest.raises(RuntimeError):                 # [.., 8, 9] + [.., 9, 8]: the reference's expression cannot broadcast
        LF.edge_aware_smoothness_per_pixel(img, pred)


@pytest.mark.gpu
def test_product_metrics_on_device(gold):
    from cc_amd import loss_functions as LF
    fns = {k: getattr(LF, k) for k in ("flow_diff", "compute_epe", "outlier_err", "compute_all_epes", "compute_errors")}
    m = {k: v.cuda() for k, v in metric_inputs().items()}
    _check(fns, m, gold, 2e-5)
