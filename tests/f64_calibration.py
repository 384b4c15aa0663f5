"""Bounds for the step-level parity tests, derived instead of hand-set.

A whole training step (21 sparse convs, ~40 train-mode BatchNorms, two chained DCNs, heads whose BatchNorm backward cancels 98 % of
its input) amplifies float32 rounding by four to five orders of magnitude: two correct float32 implementations of it differ by 1e-2 in
the gradients of the early layers.  tools/noise_locate.py follows the error module by module against a float64 run (profiles/
r05_noise_locate.txt): forward 1e-7 after the first conv, 1e-6 after the sparse backbone, x6 inside each DCN (sampling positions), 1e-4
at the head maps; backward 4e-5 at the head outputs, then x100 in the heads' first BatchNorm backward (5e-3), 2e-2 from there down.
That is the conditioning of the reference's network at these inputs; no implementation escapes it.  A bound on "HIP against the host
float32 run" can therefore only be a measured number plus a margin and moves from box to box with the host's BLAS threading (round 4's
driver run: 2.3e-2 against a hand-set 2e-2).  This file states the criterion without such numbers:

    run the same step a third time on the host in FLOAT64 (the oracle's float paths keep float64 when handed float64, oracle/__init__.py
    _work_dtype; integer outputs - rulebooks, targets, scored cells - are the float32 run's, bit for bit), and per module group require
        median_p d(hip_p, f64_p)  <=  K * median_p d(host32_p, f64_p)      (the typical parameter)
        pooled   d(hip,   f64)    <=  K * pooled   d(host32,   f64)        (all of the group's parameters as one vector)
    with d = relative L2: the HIP run may be at most K times as far from the exact gradient as the reference-style float32 host run
    is - a bound that scales by itself with the conditioning of the step at the given weights and inputs.  Where the host run happens
    to be closer to float64 than FLOOR (1e-4, north_star's per-op float tolerance), FLOOR replaces K * d.  In addition EVERY parameter
    must be within WIRING (0.25) of float64: a missing term or a wrong operand shows as O(1) in one parameter and would hide in a median.

ReLU decisions.  Where the rounding noise itself is small (the sparse backbones alone: 1e-6 at the last layers) the distance to float64 is
not a smooth quantity: profiles/r05_f64_backbone_params.txt lists it parameter by parameter - 6e-7 ... 1e-6 for BOTH float32 runs from the
output back to one layer, where it jumps to 1e-4 ... 1e-3 and stays there for everything upstream (HIP: at conv4.1.bn1.bias of the
residual backbone; the host run: one layer later, and again at conv2.2).  One pre-activation within rounding of zero has taken the
other side of its ReLU than in the float64 run; the first gradient to show it is a BatchNorm bias (a plain sum over the masked rows).
Any float32 implementation meets such events, where depends on its summation order (for the host run: on the box's BLAS threading),
so between two jump points the RATIO of the two runs' distances means nothing (x 475 ... x 1400 in that file).  The backbone test
therefore judges the whole network as one group and passes FLIPS (2e-3: the distance a few such events leave, twice the 4e-4 ... 1e-3
the HIP run shows on every box - it is run-to-run and box-to-box identical) as its floor instead of FLOOR.

K = 10 is twice the largest ratio seen: tools/step_noise.py (profiles/r05_step_noise_box*.txt: 8 weight seeds x input clouds per box,
three boxes) gives group ratios hip / host32 of 0.7 ... 5.2 for both statistics, 2.0 typical - the HIP forward pass is 1.3 ... 2.2 x
as far from float64 as the host's from the first sparse conv on (one MFMA accumulation chain over 27 offsets x channels, against
per-offset MKL products added afterwards) and the chain multiplies both alike.  The group MAXIMUM is printed but not judged: it is
heavy-tailed (0.8 ... 45 for single bias vectors the host run happens to hit within 1e-5).  The worst single HIP parameter over all
trials is 7e-2 from float64; WIRING is 3.5 x that.

Per-parameter bound (round 6, ADVICE).  A median and a norm-pooled distance cannot see a 5 - 20 % error in ONE small-norm gradient (a
BatchNorm bias, a single layer's dW), and WIRING alone only catches O(1).  So every parameter p is also held to
    d(hip_p, f64_p)  <=  max( K * d(host32_p, f64_p),  2 K * pooled d(host32, f64) of p's group,  PER_PARAM )
i.e. K times its own host distance, or twice-K times the noise level of its group at these weights and inputs (a parameter the host
run happens to hit within 1e-5 must not turn its own ratio into the verdict: the heavy tail above), or PER_PARAM = 5e-2 where both are
tiny.  Against profiles/r05_step_noise*.txt: the worst HIP parameter of the deep groups is 7.3e-2 where the group's pooled host
distance is 7.6e-3 (bound 0.15), the heads' worst 2.4e-2 (bound 5e-2): every trial of the three boxes passes with a factor >= 2, and
a single-parameter error of 5 % (shallow groups) to 15 - 20 % (deep end) now fails.  The backbone-only test passes PER_PARAM = 1e-2
(ten times the 4e-4 ... 1e-3 its parameters show on every box)."""
import statistics

K = 10.0
FLOOR = 1e-4
FLIPS = 2e-3
WIRING = 0.25
PER_PARAM = 5e-2


def rel_l2(a, truth):
    a, truth = a.detach().cpu().double(), truth.detach().cpu().double()
    return float((a - truth).norm() / truth.norm().clamp_min(1e-300))


def distances(grads, truth, skip=lambda name: False, min_norm=1e-10):
    """{name: relative L2 distance to the float64 gradient} over the parameters `truth` reaches."""
    out = {}
    for name, t in truth.items():
        if skip(name) or name not in grads or float(t.double().norm()) < min_norm:
            continue
        out[name] = rel_l2(grads[name], t)
    return out


def pooled(grads, truth, names):
    num = sum(float((grads[n].detach().cpu().double() - truth[n].detach().cpu().double()).square().sum()) for n in names)
    den = sum(float(truth[n].detach().cpu().double().square().sum()) for n in names)
    return (num / max(den, 1e-300)) ** 0.5


def compare(hip, host32, host64, group_of, skip=lambda name: False, k=K, floor=FLOOR, wiring=WIRING, per_param=PER_PARAM):
    """-> (rows, failures): rows = [(group, n, hip_max, ref_max, hip_med, ref_med, hip_pooled, ref_pooled)], failures = strings."""
    d_hip = distances(hip, host64, skip)
    d_ref = distances(host32, host64, skip)
    assert set(d_hip) == set(d_ref) and d_hip, "the three runs must reach the same parameters"
    groups = {}
    for name in d_hip:
        groups.setdefault(group_of(name), []).append(name)
    rows, bad = [], []
    for g in sorted(groups):
        names = groups[g]
        hmax, rmax = max(d_hip[n] for n in names), max(d_ref[n] for n in names)
        hmed, rmed = statistics.median(d_hip[n] for n in names), statistics.median(d_ref[n] for n in names)
        hpool, rpool = pooled(hip, host64, names), pooled(host32, host64, names)
        rows.append((g, len(names), hmax, rmax, hmed, rmed, hpool, rpool))
        if hmed > max(k * rmed, floor):
            bad.append(f"{g}: median distance {hmed:.2e} from float64 against the host float32 run's {rmed:.2e} (x{hmed / max(rmed, 1e-300):.1f} > {k})")
        if hpool > max(k * rpool, floor):
            bad.append(f"{g}: pooled distance {hpool:.2e} from float64 against the host float32 run's {rpool:.2e} (x{hpool / max(rpool, 1e-300):.1f} > {k})")
        for n in names:   # the single parameter: K x its own host distance, 2 K x the group's noise level, or the per-parameter floor
            bound = max(k * d_ref[n], 2.0 * k * rpool, per_param)
            if d_hip[n] > bound:
                bad.append(f"{n}: {d_hip[n]:.2e} from float64 against the host float32 run's {d_ref[n]:.2e}, group pooled {rpool:.2e} (bound {bound:.2e})")
    for name, d in d_hip.items():
        if d > wiring:
            bad.append(f"{name}: {d:.2e} from float64 (> {wiring}: not rounding)")
    return rows, bad


def report(rows, title="gradient distance to the float64 run, per module group"):
    r = lambda a, b: a / max(b, 1e-300)
    lines = [title, f"  {'group':42s} {'n':>3s} | {'hip med':>8s} {'h32 med':>8s} {'x':>5s} | {'hip pool':>8s} {'h32 pool':>8s} {'x':>5s} | {'hip max':>8s} {'h32 max':>8s} {'x':>5s}"]
    for g, n, hmax, rmax, hmed, rmed, hpool, rpool in rows:
        lines.append(f"  {g:42s} {n:3d} | {hmed:8.1e} {rmed:8.1e} {r(hmed, rmed):5.2f} | {hpool:8.1e} {rpool:8.1e} {r(hpool, rpool):5.2f} | {hmax:8.1e} {rmax:8.1e} {r(hmax, rmax):5.2f}")
    return "\n".join(lines)
