"""Algorithmic work of the RPN SA+FP stack, computed from shapes (SURVEY.md 8d formulas).

Used by bench.py for the roofline fractions and cross-checked in tests against the
per-scene totals quoted in BASELINE.md (85 426 176 B group_point, 113 768 448 B
three_interpolate, 13.585 GFLOP shared MLP, 71 559 936 FPS distance evaluations).
"""
from .stacks import RPN_FP, RPN_SA

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0


def rpn_stack_work(n=16384, c0=1):
    """Per-scene algorithmic totals for the b=1 RPN stack."""
    w = {"group_point_bytes": 0, "three_interpolate_bytes": 0, "mlp_flops_sa": 0, "mlp_flops_fp": 0,
         "fps_dist_evals": 0, "fps_iters": 0, "ball_query_tests": 0, "three_nn_tests": 0,
         "fused_sa_bytes": 0, "fused_fp_bytes": 0}
    npts, c = n, c0
    levels = [(n, c0)]
    for (m, radii, nss, mlps) in RPN_SA:
        w["fps_dist_evals"] += (m - 1) * npts
        w["fps_iters"] += m - 1
        cout_tot = 0
        for ns, mlp in zip(nss, mlps):
            w["ball_query_tests"] += m * npts
            w["group_point_bytes"] += m * ns * (4 + 8 * 3) + m * ns * (4 + 8 * c)
            cin = c + 3
            for co in mlp:
                w["mlp_flops_sa"] += 2 * m * ns * cin * co
                cin = co
            cout_tot += mlp[-1]
            w["fused_sa_bytes"] += 4 * npts * c + 12 * npts + 4 * m * mlp[-1]
        npts, c = m, cout_tot
        levels.append((npts, c))
    # FP: (unknown level, known level) = (3,4), (2,3), (1,2), (0,1)
    c2 = levels[4][1]
    for k, mlp in zip((3, 2, 1, 0), RPN_FP):
        n1, c1 = levels[k]
        m2 = levels[k + 1][0]
        if k == 0:
            c1 = c0 + 3
        w["three_nn_tests"] += n1 * m2
        w["three_interpolate_bytes"] += n1 * (24 + 16 * c2)
        w["fused_fp_bytes"] += 12 * n1 + 12 * m2 + n1 * 16 * c2
        cin = c2 + c1
        for co in mlp:
            w["mlp_flops_fp"] += 2 * n1 * cin * co
            cin = co
        c2 = mlp[-1]
    w["mlp_flops"] = w["mlp_flops_sa"] + w["mlp_flops_fp"]
    return w
