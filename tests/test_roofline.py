"""The roofline numerators are computed from shapes and must equal the per-scene totals
SURVEY.md 8(d) / BASELINE.md quote for the b=1 RPN stack."""
from pointrcnn_amd.roofline import rpn_stack_work


def test_rpn_stack_totals():
    w = rpn_stack_work(16384, 1)
    assert w["group_point_bytes"] == 85426176
    assert w["three_interpolate_bytes"] == 113768448
    assert w["fps_dist_evals"] == 71559936 and w["fps_iters"] == 4095 + 1023 + 255 + 63
    assert w["ball_query_tests"] == 143163392
    assert w["three_nn_tests"] == 71581696
    assert abs(w["mlp_flops_sa"] / 1e9 - 8.266) < 1e-3 and abs(w["mlp_flops_fp"] / 1e9 - 5.318) < 1e-3
