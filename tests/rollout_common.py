"""Shared logic of the rollout-parity tests: the bench workload (random ctrl in ctrlrange every 50 steps, heterogeneous envs,
Newton) through a kernel backend -- the HIP library on the GPU (`-m gpu`) or the lane emulator on the CPU -- against one fp64
oracle per env.  TEST INFRASTRUCTURE (imports oracle/).

Two protocols:
  * free-running: both sides integrate their own state for the whole rollout; reports the qpos drift of the "arm" coordinates
    (lift, arm, wrist, gripper, head) and of the "driving base" (free joint + wheel angles) separately.
  * state-synchronised: the oracle's state is copied into the kernel before every step, so each discrepancy belongs to the
    step that produced it.  A step whose acceleration differs by more than EVENT_TOL must be a BIFURCATION of the reference
    algorithm itself: the oracle, evaluated at inputs perturbed by 1e-7, has to reproduce the kernel's result.  (MPR returns
    the exit facet of the Minkowski difference; where two facets are equally close the normal jumps by tens of degrees for a
    1e-8 change of the pose -- in fp64 as well, tools/parity_probe.py.)
"""
from __future__ import annotations

import numpy as np

from oracle.oracle import Oracle

HOLD = 50
HOME = [0, 0, 0.6, 0.1, 0, 0, 0, 0, 0, 0]
EVENT_TOL = 2e-2      # relative one-step acceleration error above which a step counts as an event (a gross difference)
TYPICAL_TOL = 1.5e-3  # bound on the 99th percentile of the relative one-step acceleration error, the robot's dofs
OBJ_TOL = 5e-3        # ... and the dofs of free objects / fixture parts on their own scale: a resting object's acceleration is the small difference of
                      # gravity and a contact force of stiffness ~3e3 / s^2 and damping ~1e2 / s -- fp32 poses (1e-7 m) and velocities (1e-5) leave 1e-3 .. 3e-3 m/s^2


def ctrl_schedule(model, nu, B, windows, seed):
    cr = np.asarray(model["actuator_ctrlrange"], np.float64)
    rng = np.random.default_rng(seed)
    return [(cr[:, 0][:, None] + (cr[:, 1] - cr[:, 0])[:, None] * rng.random((nu, B))).astype(np.float32) for _ in range(windows)]


RAW_OBJ_TOL = 5e-3    # ... the same dofs when each side runs its own narrowphase: the same bound (round 6, MPR in a local frame; on world coordinates the contact points
                      # of a resting cylinder came out 0.25 mm apart in fp32 and fp64 -- 3e-2 .. 2e-1 rad/s^2 on its tilt -- which was rounding, not MPR's tolerance)
TWIN = {"manifold_keep": 1}   # oracle option: the fp64 twin of the kernels' contact-manifold cache (NOT MuJoCo; oracle/smj_oracle.c)


def settled_oracles(blob, B, solver=2, settle=500, options=None, late_options=None):
    """B oracles settled at the home keyframe from qpos0 (the bench's start).  options: {name: value} set on every oracle before the
    settling steps, late_options after them (the manifold-cache twin: the kernel starts from the settled state with an empty cache,
    so does the twin)."""
    out = []
    for _ in range(B):
        o = Oracle(blob)
        o.set_option("solver", solver)
        for k, v in (options or {}).items():
            o.set_option(k, v)
        nu = o.dim("nu")
        o.arr("ctrl")[:nu] = HOME[:nu]
        o.step(settle)
        for k, v in (late_options or {}).items():
            o.set_option(k, v)
        out.append(o)
    return out


def _fresh_oracle(blob, solver, options=None, mc=None):
    """An oracle for one evaluation at a given state; mc: the kept manifolds (Oracle.mc_export) of the oracle whose step it re-evaluates."""
    o = Oracle(blob)
    o.set_option("solver", solver)
    for k, v in (options or {}).items():
        o.set_option(k, v)
    if mc is not None:
        o.mc_import(mc)
    return o


def state_of(oracles):
    return (np.stack([o.arr("qpos") for o in oracles], 1), np.stack([o.arr("qvel") for o in oracles], 1),
            np.stack([o.arr("qacc_warmstart") for o in oracles], 1))


def drift_groups(nq):
    """Index sets of the drift report: driving base = free joint (7) + the two wheel angles; arm = every other robot
    coordinate; objects = anything beyond the robot's 27 coordinates."""
    base = list(range(0, 9))
    arm = list(range(9, 27))
    obj = list(range(27, nq))
    return base, arm, obj


def free_running(backend, blob, model, B, windows, seed, solver=2, band=1e-4, kernel_keeps_manifolds=True):
    """Returns per-env max drift over the rollout: dict(base=[B], arm=[B], obj=[B]), the per-window history and the flags -- and, for
    every env whose drift leaves `band` IN ANY COORDINATE (robot or objects), WHY (`departures`, see explain_departures): the rollout
    keeps the kernel's and the oracles' state at every window boundary so that the stretch in which an env leaves can be run again step
    by step afterwards.  The oracles of the rollout are the UNMODIFIED restatement (north_star's comparison); the step-by-step replay
    that explains a departure evaluates the kernel's steps against the oracle's twin of the manifold cache when the kernel runs with
    it (kernel_keeps_manifolds; the rule itself is bounded oracle against oracle, tests/test_satellites.py)."""
    oracles = settled_oracles(blob, B, solver)
    nq, nu = oracles[0].dim("nq"), oracles[0].dim("nu")
    backend.upload(*state_of(oracles))
    sched = ctrl_schedule(model, nu, B, windows, seed)
    base, arm, obj = drift_groups(nq)
    hist = []
    snaps_k, snaps_o = [], []          # state at the START of window w (entry `windows`: the end): the kernel's (fp32), the oracles' (fp64)
    for w in range(windows):
        k = backend.download()
        snaps_k.append((k["qpos"].copy(), k["qvel"].copy(), k["warm"].copy()))
        snaps_o.append(state_of(oracles))
        backend.set_ctrl(sched[w])
        for b, o in enumerate(oracles):
            o.arr("ctrl")[:nu] = sched[w][:, b]
            o.step(HOLD)
        backend.step(HOLD)
        q = backend.download()["qpos"]
        d = np.abs(q - np.stack([o.arr("qpos") for o in oracles], 1))
        hist.append((d[base].max(0), d[arm].max(0), d[obj].max(0) if obj else np.zeros(B)))
    snaps_o.append(state_of(oracles))
    mx = lambda k: np.max(np.stack([h[k] for h in hist]), 0)
    flags = backend.download()["info"][3].copy()
    robot = np.stack([np.maximum(np.maximum(h[0], h[1]), h[2]) for h in hist])          # [windows, B], every coordinate
    left = {b: int(np.argmax(robot[:, b] >= band)) for b in range(B) if (robot[:, b] >= band).any()}
    keeps = kernel_keeps_manifolds and int(np.asarray(model["dims"]).ravel()[1]) > 32   # (the kernels' default: off for a robot alone, smj_model_load.h)
    departures = explain_departures(backend, blob, model, sched, snaps_k, snaps_o, left, solver, band, oracle_options=TWIN if keeps else None)
    return dict(base=mx(0), arm=mx(1), obj=mx(2), hist=hist, flags=flags, oracles=oracles, departures=departures)


def step_error_parts(qk, qa, nrobot=26):
    """(robot, objects): one-step acceleration error, relative, the robot's dofs and everything else (free objects, fixture parts) each
    on their own scale; objects = 0 where the scene has none."""
    e = np.abs(qk - qa)
    r = e[:nrobot].max() / max(1.0, np.abs(qa[:nrobot]).max())
    ob = e[nrobot:].max() / max(1.0, np.abs(qa[nrobot:]).max()) if len(qa) > nrobot else 0.0
    return float(r), float(ob)


def step_error(qk, qa, nrobot=26):
    """THE metric of every one-step comparison, detector and explainers alike: the larger of the two parts above -- a finger joint at
    1e4 rad/s^2 must not hide a gross error of a resting object, nor the other way round."""
    return max(step_error_parts(qk, qa, nrobot))


def _sensitivity_explains(blob, solver, state, ctrl, qk, qa, trials=8, options=None, mc=None):
    """A step whose error is small on the scale of EVENT_TOL but large on the scale of the drift band (violent contact phases: 1e3-1e5
    rad/s^2, where 1e-3 relative is 50 steps' worth of the band): is the kernel's deviation within what the reference algorithm itself
    does with an input perturbed at fp32 resolution?  The oracle is run at inputs perturbed by 1e-7 (fp32 rounding of the poses), 1e-6
    (MPR's own tolerance) and 1e-5; explained when a perturbed run reproduces at least half of the kernel's deviation, or when the
    perturbed runs scatter around the unperturbed one by at least the kernel's deviation.  Returns (explained, eps, spread / error)."""
    qpos, qvel, warm = state
    nr = min(26, len(qa))
    wgt = np.full(len(qa), 1.0 / max(1.0, np.abs(qa[:nr]).max()))
    if len(qa) > nr:
        wgt[nr:] = 1.0 / max(1.0, np.abs(qa[nr:]).max())
    err = float((wgt * np.abs(qk - qa)).max())
    rng = np.random.default_rng(4321)
    spread = 0.0
    for eps in (1e-7, 1e-6, 1e-5):
        for t in range(trials):
            o = _fresh_oracle(blob, solver, options, mc)
            o.arr("qpos")[:] = qpos + rng.normal(size=qpos.shape) * eps
            o.arr("qvel")[:] = qvel; o.arr("qacc_warmstart")[:] = warm
            o.arr("ctrl")[: len(ctrl)] = ctrl
            o.forward()
            qp = o.arr("qacc").copy()
            o.close()
            spread = max(spread, float((wgt * np.abs(qp - qa)).max()))
            if float((wgt * np.abs(qk - qp)).max()) <= 0.5 * err or spread >= err:
                return True, eps, spread / max(err, 1e-30)
    return False, None, spread / max(err, 1e-30)


def explain_departures(backend, blob, model, sched, snaps_k, snaps_o, left, solver, band, oracle_options=None):
    """Every env that leaves the drift band must say why.  For env b leaving in window w the kernel is run again from ITS OWN state at
    the start of window w - 1 (where both sides were still inside the band), one step at a time, and at every step the oracle is
    evaluated on the kernel's state of that step (state-synchronised along the kernel's trajectory: each discrepancy belongs to the step
    that produced it).  The departure is explained as

      * "bifurcation": at least one of those <= 100 steps is an EVENT -- a one-step acceleration error above EVENT_TOL (step_error: robot
        dofs and object dofs each on their own scale) or a different contact count -- and every event is one the reference algorithm
        itself produces: the oracle reproduces the kernel's result, IN THE SAME METRIC, at an input PERTURBED by 1e-7 .. 1e-5 (never the
        unperturbed evaluation: that is the one that raised the event), or on the kernel's contact list (see state_synchronised); or
      * "sensitivity": no gross event, but steps whose absolute error matters on the scale of the band (more than band / (2 HOLD dt^2) =
        0.25 rad/s^2 on a robot dof: the violent phases, fingers against objects at 1e3-1e5 rad/s^2), and the largest of them (up to 10)
        are each within the reference algorithm's own response to an input perturbed at fp32 resolution (_sensitivity_explains); or
      * "conditioning": there is no event -- every step the kernel took is the oracle's step to within the one-step tolerance -- and
        the trajectory is not determined to within the band by its state at the start of the stretch: either two fp64 oracles, one from
        the oracle's state and one from the kernel's in-band state at the start of window w - 1, end >= band / 2 apart, or the kernel's
        second run from the same state does not leave the band at all (the two runs differ only by what the contact-manifold cache
        holds: reuse within 3e-7 of a pose).

    Anything else is "unexplained" and fails the test.  Returns {env: dict(window, kind, events, ...)}.

    oracle_options: options of the replay's oracles -- TWIN when the kernel keeps manifolds (its default): each env's replay oracle then
    lives through the whole stretch, keeps manifolds by the kernel's rule, and hands them (mc_export) to the oracles that re-evaluate
    one of its steps; the kernel's own kept manifolds are dropped before the replay starts (backend.clear_caches), as the twin's are."""
    out = {}
    if not left:
        return out
    twin = bool(oracle_options and oracle_options.get("manifold_keep"))
    nu = sched[0].shape[0]
    envs = sorted(left)
    w0 = {b: max(left[b] - 1, 0) for b in envs}
    nwin = {b: left[b] - w0[b] + 1 for b in envs}
    # the kernel's second run: slot b runs env b's stretch; the slots of envs that stayed inside idle at their last state
    qpos, qvel, warm = (a.copy() for a in snaps_k[-1])
    for b in envs:
        for dst, src in zip((qpos, qvel, warm), snaps_k[w0[b]]):
            dst[:, b] = src[:, b]
    backend.clear_caches()
    backend.upload(qpos, qvel, warm)
    ctrl = sched[-1].copy()
    events = {b: [] for b in envs}
    steprel = {b: [] for b in envs}
    relevant = {b: [] for b in envs}     # steps whose absolute error matters for the band: (error, step, state, ctrl, qk, qa)
    dt = float(np.asarray(model["opt_timestep"]).ravel()[0]) if "opt_timestep" in model else 0.002
    a_band = band / (2 * HOLD * dt * dt)
    endq = {}
    o = {b: _fresh_oracle(blob, solver, oracle_options) for b in envs}
    nv = o[envs[0]].dim("nv")
    for s in range(2 * HOLD):
        for b in envs:
            if s < nwin[b] * HOLD:
                ctrl[:, b] = sched[w0[b] + s // HOLD][:, b]
        backend.set_ctrl(ctrl)
        pre = backend.download()
        backend.upload(pre["qpos"], pre["qvel"], pre["warm"])      # an exact round trip; lets the emulator backend hand a step over like smj_step
        backend.step(1)
        post = backend.download()
        for b in envs:
            if s >= nwin[b] * HOLD:
                continue
            st = (pre["qpos"][:, b].astype(np.float64), pre["qvel"][:, b].astype(np.float64), pre["warm"][:, b].astype(np.float64))
            ob = o[b]
            ob.arr("qpos")[:] = st[0]; ob.arr("qvel")[:] = st[1]; ob.arr("qacc_warmstart")[:] = st[2]
            ob.arr("ctrl")[:nu] = ctrl[:, b]
            mc = ob.mc_export() if twin else None      # the manifolds the step starts with
            ob.step(1)
            qa, qk = ob.arr("qacc").copy(), post["qacc"][:nv, b]
            r = step_error(qk, qa)
            steprel[b].append(r)
            nk = int(post["info"][1, b])
            if r > EVENT_TOL or nk != ob.ncon:
                ok = False
                if nk == ob.ncon:   # (cheap: an event that is gone on the kernel's own contact list belongs to the narrowphase, not to the dynamics)
                    ok, err = _same_contacts_same_dynamics(blob, solver, st, ctrl[:, b], qk, post["contacts"][:, b], nk)
                    eps = "kernel contacts" if ok else None
                if not ok:
                    ok, err, eps = _bifurcation_explains(blob, solver, st, ctrl[:, b], qk, options=oracle_options, mc=mc)
                if not ok and nk != ob.ncon:
                    ok, err = _same_contacts_same_dynamics(blob, solver, st, ctrl[:, b], qk, post["contacts"][:, b], nk)
                    eps = "kernel contacts" if ok else None
                events[b].append(dict(step=w0[b] * HOLD + s, rel=r, ncon_kernel=nk, ncon_oracle=ob.ncon, explained=bool(ok),
                                      residual=float(err), eps=eps, flags=int(post["info"][3, b])))
            ea = float(np.abs(qk - qa).max())
            if ea > a_band:
                relevant[b].append((ea, w0[b] * HOLD + s, st, ctrl[:, b].copy(), qk.copy(), qa, mc))
            if s == nwin[b] * HOLD - 1:
                endq[b] = post["qpos"][:, b].copy()
    base, arm, objc = drift_groups(snaps_k[0][0].shape[0])
    robot = base + arm + objc     # (every coordinate: an env may leave the band through an object)
    for b in envs:
        ev = events[b]
        d0 = float(np.abs(snaps_k[w0[b]][0][robot, b] - snaps_o[w0[b]][0][robot, b]).max())
        again = float(np.abs(endq[b][robot] - snaps_o[left[b] + 1][0][robot, b]).max())     # the second run against the free-running oracle
        rec = dict(window=left[b], events=ev, start_drift=d0, step_rel_p99=float(np.percentile(steprel[b], 99)), second_run_drift=again,
                   amplification=None, oracle_pair_drift=None)
        rec["relevant_steps"] = len(relevant[b])
        if ev:
            rec["kind"] = "bifurcation" if all(e["explained"] for e in ev) else "unexplained"
        elif relevant[b]:
            checks = []
            for ea, step, st, c, qk, qa, mc in sorted(relevant[b], key=lambda t: -t[0])[:10]:
                ok, eps, ratio = _sensitivity_explains(blob, solver, st, c, qk, qa, options=oracle_options, mc=mc)
                checks.append(dict(step=step, abs_err=ea, explained=ok, eps=eps, spread_over_err=ratio))
            rec["sensitivity"] = checks
            rec["kind"] = "sensitivity" if all(c["explained"] for c in checks) else "unexplained"
        else:
            pair = []
            for snap in (snaps_o, snaps_k):
                ob = Oracle(blob)
                ob.set_option("solver", solver)
                ob.arr("qpos")[:] = snap[w0[b]][0][:, b]; ob.arr("qvel")[:] = snap[w0[b]][1][:, b]
                ob.arr("qacc_warmstart")[:] = snap[w0[b]][2][:, b]
                for w in range(w0[b], left[b] + 1):
                    ob.arr("ctrl")[:nu] = sched[w][:, b]
                    ob.step(HOLD)
                pair.append(ob.arr("qpos")[robot].copy())
                ob.close()
            d1 = float(np.abs(pair[0] - pair[1]).max())
            rec["amplification"] = d1 / max(d0, 1e-12)
            rec["oracle_pair_drift"] = d1
            clean = rec["step_rel_p99"] < 4 * TYPICAL_TOL
            rec["kind"] = "conditioning" if clean and (d1 >= 0.5 * band or again < band) else "unexplained"
        out[b] = rec
    return out


def _bifurcation_explains(blob, solver, state, ctrl, qacc_kernel, trials=12, tol=EVENT_TOL, options=None, mc=None):
    """Does the oracle reproduce the kernel's acceleration at an input NEAR the shared state?  Perturbations of 1e-7 (fp32
    round-off of the poses), then 1e-6 and 1e-5: MPR's answer on curved or faceted pairs (cylinder rim against a mesh hull) is
    the normal of the portal facet it happens to stop on, and which facet that is moves with perturbations far below the
    algorithm's own 1e-6 m tolerance.  The error is step_error -- the metric that raised the event: robot and object dofs each on
    their own scale -- and every trial is perturbed: the unperturbed evaluation is the one the event was raised against (VERDICT r5
    "weak" 1: with the whole-vector scale and an unperturbed trial 0 every object-dof event was "explained" by that very evaluation).
    Returns (explained, best residual, eps that explained it)."""
    qpos, qvel, warm = state
    rng = np.random.default_rng(12345)
    best = np.inf
    for eps in (1e-7, 1e-6, 1e-5):
        for t in range(trials):
            o = _fresh_oracle(blob, solver, options, mc)
            o.arr("qpos")[:] = qpos + rng.normal(size=qpos.shape) * eps
            o.arr("qvel")[:] = qvel; o.arr("qacc_warmstart")[:] = warm
            o.arr("ctrl")[: len(ctrl)] = ctrl
            o.forward()
            err = step_error(qacc_kernel, o.arr("qacc"))
            o.close()
            best = min(best, err)
            if err < tol:
                return True, err, eps
    return False, best, None


def _kernel_contacts(dump, ncon_k):
    """The kernel's contact list (debug slot) in the layout of Oracle.set_contacts: dist, pos3, normal3, geom1, geom2."""
    ck = dump.reshape(-1, 8)[:ncon_k].astype(np.float64)
    code = ck[:, 7].astype(np.int64)
    return np.concatenate([ck[:, :7], ((code >> 4) & 1023)[:, None].astype(np.float64), (code >> 14)[:, None].astype(np.float64)], 1)


def _same_contacts_same_dynamics(blob, solver, state, ctrl, qacc_kernel, dump, ncon_k, tol=EVENT_TOL):
    qpos, qvel, warm = state
    con = _kernel_contacts(dump, ncon_k)
    o = Oracle(blob)
    o.set_option("solver", solver)
    o.arr("qpos")[:] = qpos; o.arr("qvel")[:] = qvel; o.arr("qacc_warmstart")[:] = warm
    o.arr("ctrl")[: len(ctrl)] = ctrl
    o.set_contacts(con)
    o.forward()
    err = step_error(qacc_kernel, o.arr("qacc"))     # (the detector's metric)
    o.close()
    return bool(err < tol), float(err)


def state_synchronised(backend, blob, model, B, windows, seed, solver=2, oracle_options=None, twin=False):
    """Per-step comparison on identical inputs.  Returns (relative qacc errors [steps*B] in the metric step_error: robot and object dofs
    each on their own scale, the larger of the two; the parts are left in state_synchronised.rel_robot / .rel_obj), events) where every
    event is a dict with the step, the error and whether a perturbation of the oracle's input reproduces the kernel's result.
    twin: the oracles keep contact manifolds by the kernels' rule from the settled state on (the kernel must run with its manifold
    cache on): the comparison is then of the IMPLEMENTATION of that rule; without it (kernel: manifold_cache 0) of the dynamics."""
    oracles = settled_oracles(blob, B, solver, options=oracle_options, late_options=TWIN if twin else None)
    xopt = dict(oracle_options or {}, **(TWIN if twin else {}))
    rel_robot, rel_obj = [], []
    # the DYNAMICS on identical contacts, every env-step: a second oracle per env is handed the kernel's contact list at the shared state --
    # separates what the narrowphase contributes to a one-step error (contact points, depths, normals in fp32) from what the solver does
    shadow = [_fresh_oracle(blob, solver, oracle_options) for _ in range(B)]
    same_robot, same_obj = [], []
    nu, nv = oracles[0].dim("nu"), oracles[0].dim("nv")
    sched = ctrl_schedule(model, nu, B, windows, seed)
    rel, events = [], []
    clean = []    # env-steps whose contact lists agree (same pairs, normals within 0.5 degree): the dynamics-only error sample
    iters = []    # per env-step: (solver iterations of the kernel, of the oracle)
    cstat = dict(n=0, depth=[], pos=[], cosn=[], mismatched_steps=0)   # contact geometry on identical states
    for w in range(windows):
        backend.set_ctrl(sched[w])
        for b, o in enumerate(oracles):
            o.arr("ctrl")[:nu] = sched[w][:, b]
        for s in range(HOLD):
            st = state_of(oracles)
            backend.upload(*st)
            backend.step(1)
            out = backend.download()
            for b, o in enumerate(oracles):
                mc = o.mc_export() if twin else None
                o.step(1)
                qa = o.arr("qacc")
                qk = out["qacc"][:nv, b]
                rr, ro = step_error_parts(qk, qa)
                r = max(rr, ro)
                rel.append(r); rel_robot.append(rr); rel_obj.append(ro)
                sh = shadow[b]
                sh.arr("qpos")[:] = st[0][:, b]; sh.arr("qvel")[:] = st[1][:, b]; sh.arr("qacc_warmstart")[:] = st[2][:, b]
                sh.arr("ctrl")[:nu] = sched[w][:, b]
                sh.set_contacts(_kernel_contacts(out["contacts"][:, b], int(out["info"][1, b])))
                sh.forward()
                sr, so = step_error_parts(qk, sh.arr("qacc"))
                same_robot.append(sr); same_obj.append(so)
                iters.append((int(out["info"][2, b]), int(o.iarr("solver_niter")[0])))
                if _compare_contacts(cstat, out["contacts"][:, b], int(out["info"][1, b]), o):
                    clean.append((r, rr, ro))
                if r > EVENT_TOL or int(out["info"][1, b]) != o.ncon:
                    if max(sr, so) < EVENT_TOL and int(out["info"][1, b]) == o.ncon and rr <= EVENT_TOL:
                        # an object-dof event with the oracle's pair list, gone on the kernel's contact list: the two narrowphases' contact points differ
                        ok, err, eps = True, max(sr, so), "contact-point scatter"
                    else:
                        ok, err, eps = _bifurcation_explains(blob, solver, (st[0][:, b], st[1][:, b], st[2][:, b]), sched[w][:, b], qk, options=xopt, mc=mc)
                    if not ok:
                        # same state, the KERNEL's contact list handed to the oracle: isolates the dynamics from MPR's portal
                        # noise on curved rims / faceted hulls (its normal is the facet the refinement stops on: degrees of
                        # scatter under the algorithm's own 1e-6 m tolerance, tools/parity_probe.py)
                        ok, err = _same_contacts_same_dynamics(blob, solver, (st[0][:, b], st[1][:, b], st[2][:, b]), sched[w][:, b], qk,
                                                               out["contacts"][:, b], int(out["info"][1, b]))
                        eps = "kernel contacts" if ok else None
                    events.append(dict(env=b, window=w, step=s, rel=float(r), ncon_kernel=int(out["info"][1, b]), ncon_oracle=o.ncon,
                                       explained=bool(ok), residual=float(err), eps=eps, flags=int(out["info"][3, b])))
    state_synchronised.contacts = cstat
    cl = np.array(clean).reshape(-1, 3)
    state_synchronised.clean, state_synchronised.clean_robot, state_synchronised.clean_obj = cl[:, 0], cl[:, 1], cl[:, 2]
    state_synchronised.iters = np.array(iters)
    state_synchronised.rel_robot, state_synchronised.rel_obj = np.array(rel_robot), np.array(rel_obj)
    state_synchronised.same_robot, state_synchronised.same_obj = np.array(same_robot), np.array(same_obj)
    return np.array(rel), events


def gross_events(events):
    """Events other than contact-point scatter (an object-dof error that is gone on the kernel's contact list, pair lists equal: rare since the
    narrowphase runs in a local frame)."""
    return [ev for ev in events if ev["eps"] != "contact-point scatter"]


def assert_object_dofs(tag=""):
    """The object-dof part of the last state_synchronised run, printed and bounded on their own scale, every step: on identical contacts
    (OBJ_TOL) and on each side's own narrowphase (RAW_OBJ_TOL, the same value)."""
    S = state_synchronised
    pc = lambda a: f"p50 {np.percentile(a, 50):.1e} p99 {np.percentile(a, 99):.1e} max {a.max():.1e}"
    print(f"   {tag}object / fixture dofs on their own scale: own narrowphase {pc(S.rel_obj)}; on the kernel's contact list {pc(S.same_obj)} (robot dofs there: {pc(S.same_robot)})")
    assert np.percentile(S.same_obj, 99) < OBJ_TOL and np.percentile(S.same_robot, 99) < TYPICAL_TOL * 4, (pc(S.same_obj), pc(S.same_robot))
    assert np.percentile(S.rel_obj, 99) < RAW_OBJ_TOL, pc(S.rel_obj)


def _compare_contacts(cstat, dump, ncon_k, o):
    """Contact list of the kernel (debug slot: dist, pos, normal, condim | geom1 << 4 | geom2 << 14 per contact) against the
    oracle's on the same state, contact by contact.  Both lists are put in geom-pair order first (stable: the contacts of one pair
    keep their order): the oracle emits in pair-table order, the kernel plane pairs first, then the pairs with static geoms, then
    the moving-moving pairs, each group in table order."""
    n = o.ncon
    ck = dump.reshape(-1, 8)[:ncon_k]
    co = o.arr("contact").reshape(n, -1) if n else np.zeros((0, 29))
    code = ck[:, 7].astype(np.int64)
    gk = [(int((c >> 4) & 1023), int(c >> 14)) for c in code]
    go = [tuple(int(v) for v in co[k, -2:].copy().view(np.int32)[1:3]) for k in range(n)]
    ik, io = sorted(range(len(gk)), key=lambda k: gk[k]), sorted(range(n), key=lambda k: go[k])
    gk, go, ck, co = [gk[k] for k in ik], [go[k] for k in io], ck[ik], co[io]
    if gk != go:
        cstat["mismatched_steps"] += 1
        return False
    same = True
    for k in range(n):
        same = same and float(np.dot(ck[k, 4:7], co[k, 4:7])) > 0.99996 and abs(ck[k, 0] - co[k, 0]) < 1e-5
        cstat["n"] += 1
        cstat["depth"].append(abs(ck[k, 0] - co[k, 0]))
        cstat["pos"].append(np.abs(ck[k, 1:4] - co[k, 1:4]).max())
        cstat["cosn"].append(float(np.dot(ck[k, 4:7], co[k, 4:7])))
    return same


class EmulBackend:
    """The kernel source through the CPU lane emulator (tests/emul).  The emulator has no escalation of its own; for single steps
    from an uploaded state (the state-synchronised protocol) this backend does what smj_step does on the device: an env whose step
    ran out of constraint rows / contacts in the primary variant is stepped again, from the same state, by the variant the
    device hands it to (standard / mid -> tall, big38 / big50 -> big, sat -> sat32), and that result is the one reported."""
    ESC = {"standard": "tall", "mid": "tall", "big38": "big", "big50": "big", "sat": "sat32"}

    def __init__(self, blob, B, solver=2, variant=None):
        from emul.emul import Emul

        o = Oracle(blob)
        self.dims = dict(nq=o.dim("nq"), nv=o.dim("nv"), nu=o.dim("nu"), nlidar=360)
        self.blob, self.B = blob, B
        self.e = Emul(blob, self.dims, num_envs=B, debug=True, variant=variant)   # variant as smj_create picks it
        self._opts = {}
        orig = self.e.set_option

        def record(name, v):   # options set on the primary reach the hand-over variant too
            self._opts[name] = v
            return orig(name, v)
        self.e.set_option = record
        self.e.set_option("solver", solver)
        from stretch_mujoco_amd.lib import debug_layout
        import stretch_mujoco_amd.model_blob as mb

        self.D = debug_layout(self.e.nvp, self.e.ncon_max, self.e.nsat_max)
        self.ncon_max, self.nvp = self.e.ncon_max, self.e.nvp
        self.model = mb.loads(blob)
        self.x = None          # the hand-over variant's emulator, made when first needed
        self.handed = []       # envs of the last step that it finished
        self._state = None

    def upload(self, qpos, qvel, warm):
        self.e.qpos[:] = qpos; self.e.qvel[:] = qvel; self.e.warm[:] = warm
        self._state = (np.array(qpos, np.float32), np.array(qvel, np.float32), np.array(warm, np.float32))

    def set_ctrl(self, ctrl):
        self.e.ctrl[:] = ctrl

    def clear_caches(self):
        """Drop what the kernel keeps between steps beside the state (kept manifolds, separating directions, the PGS second start)."""
        self.e.clear_caches()
        if self.x is not None:
            self.x.clear_caches()

    def step(self, n):
        self.handed = []
        st, self._state = self._state, None
        if n != 1 or st is None or self.e.variant not in self.ESC:
            self.e.step(n)
            return
        ctrl0 = self.e.ctrl.copy()
        before = self.e.info[3].copy()
        self.e.info[3] = 0
        self.e.step(1)
        over = [b for b in range(self.B) if int(self.e.info[3, b]) & 3]
        if over:
            from emul.emul import Emul
            from stretch_mujoco_amd.lib import debug_layout

            if self.x is None:
                self.x = Emul(self.blob, self.dims, num_envs=self.B, debug=True, variant=self.ESC[self.e.variant])
                self.xD = debug_layout(self.x.nvp, self.x.ncon_max, self.x.nsat_max)
            for k, v in self._opts.items():
                self.x.set_option(k, v)
            self.x.qpos[:] = st[0]; self.x.qvel[:] = st[1]; self.x.warm[:] = st[2]; self.x.ctrl[:] = ctrl0
            self.x.info[3] = 0
            self.x.step(1)
            for b in over:
                self.e.qpos[:, b] = self.x.qpos[:, b]; self.e.qvel[:, b] = self.x.qvel[:, b]; self.e.warm[:, b] = self.x.warm[:, b]
                self.e.info[:, b] = self.x.info[:, b]
            self.handed = over
        self.e.info[3] |= before

    def download(self):
        e = self.e
        from stretch_mujoco_amd.lib import full_qacc

        def parts(em, D):
            qacc = full_qacc(em.debug, D, self.model) if em.nsat_max else em.debug[D["qacc"]:D["qacc"] + em.nvp]
            return qacc.astype(np.float64), em.debug[D["con"]:D["con"] + 8 * em.ncon_max].copy()

        qacc, con = parts(e, self.D)
        if self.handed:
            qx, cx = parts(self.x, self.xD)
            big = np.zeros((max(con.shape[0], cx.shape[0]), self.B), con.dtype)
            big[:con.shape[0]] = con
            nq = min(qacc.shape[0], qx.shape[0])
            for b in self.handed:
                qacc[:nq, b] = qx[:nq, b]
                big[:, b] = 0; big[:cx.shape[0], b] = cx[:, b]
            con = big
        return dict(qpos=e.qpos.astype(np.float64), qvel=e.qvel.astype(np.float64), warm=e.warm.astype(np.float64), info=e.info.copy(), qacc=qacc,
                    contacts=con)


class HipBackend:
    """libsmj.so through StretchBatchSimulator (the C-ABI path)."""

    def __init__(self, scene, B, solver=2):
        import torch

        from stretch_mujoco_amd import StretchBatchSimulator

        self.torch = torch
        self.sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene=scene, solver={0: "pgs", 2: "newton"}[solver], debug=True)
        self.sim.start(home=False)
        self.sim.set_option("pgs_dual_warmstart", 0)   # MuJoCo's own warm start, as the unmodified oracle's; tests of the default (1) set it, on both sides
        self.D, self.nvp, self.ncon_max = self.sim.debug_layout, self.sim.nv_max, self.sim.ncon_max

    def _put(self, dst, a):
        dst.copy_(self.torch.as_tensor(np.ascontiguousarray(a), dtype=self.torch.float32).to(dst.device))

    def upload(self, qpos, qvel, warm):
        self._put(self.sim.qpos, qpos); self._put(self.sim.qvel, qvel); self._put(self.sim.qacc_warmstart, warm)

    def set_ctrl(self, ctrl):
        self._put(self.sim.ctrl, ctrl)

    def clear_caches(self):
        """smj_reset drops what the library keeps between steps beside the state (the callers upload a state next)."""
        self.sim.reset()

    def step(self, n):
        self.sim.step(n)

    def download(self):
        s = self.sim
        self.torch.cuda.synchronize()
        from stretch_mujoco_amd.lib import full_qacc

        dbg = s.debug.cpu().numpy()
        qacc = full_qacc(dbg, self.D, s.model) if s.nsat_max else dbg[self.D["qacc"]:self.D["qacc"] + self.nvp]
        return dict(qpos=s.qpos.cpu().numpy().astype(np.float64), qvel=s.qvel.cpu().numpy().astype(np.float64),
                    warm=s.qacc_warmstart.cpu().numpy().astype(np.float64), info=s.info.cpu().numpy(), qacc=qacc.astype(np.float64),
                    contacts=s.debug[self.D["con"]:self.D["con"] + 8 * self.ncon_max].cpu().numpy())

    def close(self):
        self.sim.stop()
