"""The BENCH workload (heterogeneous random ctrl every 50 steps, Newton) against the fp64 oracle, env by env: through the lane
emulator on the CPU and through libsmj.so on the GPU (`-m gpu`).  Drift is reported separately for the arm coordinates and the
driving base; one-step discrepancies above EVENT_TOL must be bifurcations of the reference algorithm itself
(rollout_common.py)."""
import numpy as np
import pytest

import rollout_common as rc
import stretch_mujoco_amd.model_blob as mb
from conftest import MODELS

SCENES = ["stretch_empty", "stretch_kitchen_standin", "stretch_scene", "stretch_kitchen4"]


def _blob(scene):
    with open(f"{MODELS}/{scene}.smjb", "rb") as f:
        b = f.read()
    return b, mb.loads(b)


def _keeps_manifolds(model):
    """The kernels' default for the manifold cache (smj_model_load.h): on for models of more than 32 dofs -- a robot alone has no resting convex pair."""
    return int(np.asarray(model["dims"]).ravel()[1]) > 32


def _report(tag, r):
    print(f"\n[{tag}] max qpos drift over the rollout, per env:")
    print("   driving base:", np.array2string(r["base"], precision=1, floatmode="fixed", formatter={"float_kind": lambda v: f"{v:.1e}"}))
    print("   arm         :", np.array2string(r["arm"], formatter={"float_kind": lambda v: f"{v:.1e}"}))
    if r["obj"].any():
        print("   free object :", np.array2string(r["obj"], formatter={"float_kind": lambda v: f"{v:.1e}"}))


OBJECT_SCENES = ("stretch_scene", "stretch_kitchen4", "stretch_kitchen_robocasa")   # free objects (and a table) within the arm's reach

# north_star: qpos drift < 1e-4 over 1000 steps.  The bound is asserted PER ENV with a cause (round 5; rounds 3-4 asserted "the measured
# fraction of envs minus one", which a regression of one env per scene would have passed): an env may leave the band only through an
# EXPLAINED departure -- rollout_common.explain_departures replays the stretch in which it leaves, state-synchronised along the kernel's
# own trajectory, and accepts (i) a bifurcation of the reference algorithm itself (the oracle reproduces the kernel's step at an input
# perturbed by <= 1e-5, or on the kernel's contact list), (ii) sensitivity (violent phases: the steps whose error matters for the band are each
# within the oracle's own response to an input perturbed at fp32 resolution) or (iii) conditioning (no such step at all, and two fp64 oracles
# started from the two in-band states end further apart than half the band).  The floors below only keep the test from passing vacuously.
FLOOR_INSIDE = {"stretch_empty": 0.5, "stretch_kitchen_standin": 0.5, "stretch_scene": 0.0, "stretch_kitchen4": 0.25, "stretch_kitchen_robocasa": 0.25}   # (fixed numbers, not "the last measurement minus one": round 6 measures 9-10 of 16 in the Robocasa-scale kitchen after the fp32 contact-normal fix)
FLOOR_INSIDE_EARLY = 0.7     # scenes with free objects in reach: the first 250 steps, the robot's coordinates (the criterion of rounds 3-5, kept)
FLOOR_INSIDE_EARLY_ALL = 0.5 # ... and every coordinate, the objects' included (round 6: the stronger verdict; objects the gripper is already pushing leave first)
OBJ_BAND = 1e-4              # qpos of free objects / fixture parts: the same band as the robot's coordinates


def _check_free_running(r, scene):
    B = len(r["base"])
    assert (r["flags"] == 0).all(), r["flags"]
    # the verdict is over EVERY coordinate: north_star says "per-env qpos ... match", and in the kitchens most of qpos is objects / fixture parts
    ok = (r["base"] < 1e-4) & (r["arm"] < 1e-4) & (r["obj"] < OBJ_BAND)
    print(f"   inside 1e-4 over the whole rollout (robot and objects): {int(ok.sum())} / {B}; robot alone: {int(((r['base'] < 1e-4) & (r['arm'] < 1e-4)).sum())} / {B}")
    dep = r["departures"]
    assert set(dep) == {b for b in range(B) if not ok[b]}
    for b in sorted(dep):
        d = dep[b]
        ev = d["events"]
        if ev:
            what = (f"{len(ev)} event(s), first at step {ev[0]['step']}: rel {ev[0]['rel']:.1e}, contacts {ev[0]['ncon_kernel']} / {ev[0]['ncon_oracle']}, "
                    f"explained by {ev[0]['eps']}")
        elif d.get("sensitivity"):
            c = d["sensitivity"]
            what = (f"no gross step; {d['relevant_steps']} steps with an error that matters for the band (largest {c[0]['abs_err']:.1e} rad/s^2 at step "
                    f"{c[0]['step']}), the {len(c)} largest each within the oracle's own response to a perturbation of {max(x['eps'] or 0 for x in c):.0e}"
                    if all(x["explained"] for x in c) else f"steps not explained by sensitivity: {[x for x in c if not x['explained']]}")
        else:
            what = (f"no gross step (p99 one-step error {d['step_rel_p99']:.1e}); two fp64 oracles from the two in-band states ({d['start_drift']:.1e} "
                    f"apart) end {d['oracle_pair_drift']:.1e} apart; the kernel's second run from the same state ends {d['second_run_drift']:.1e} from the oracle")
        print(f"   env {b} leaves in window {d['window']} (steps {50 * d['window']}..{50 * d['window'] + 49}): {d['kind']}: {what}")
    bad = {b: d for b, d in dep.items() if d["kind"] == "unexplained"}
    assert not bad, bad
    assert all(e["flags"] == 0 for d in dep.values() for e in d["events"])
    if scene in OBJECT_SCENES:
        # Manipulation of 0.1-0.5 kg objects is chaotic: once the gripper has knocked one over, the fp32 and fp64 runs are two
        # different rollouts (MPR's portal noise on cylinder rims seeds it, tools/parity_probe.py) -- as two MuJoCo builds would be.
        h = r["hist"][:5]
        early = np.max(np.stack([np.maximum(x[0], x[1]) for x in h]), 0)
        early_all = np.max(np.stack([np.maximum(np.maximum(x[0], x[1]), x[2]) for x in h]), 0)
        print(f"   inside 1e-4 over the first 250 steps: robot {int((early < 1e-4).sum())} / {B}, robot and objects {int((early_all < 1e-4).sum())} / {B}")
        assert (early < 1e-4).mean() >= FLOOR_INSIDE_EARLY, early
        assert (early_all < 1e-4).mean() >= FLOOR_INSIDE_EARLY_ALL, early_all
    assert ok.mean() >= FLOOR_INSIDE[scene], (ok.mean(), r["base"], r["arm"], r["obj"])
    if scene not in OBJECT_SCENES:
        assert np.median(np.maximum(np.maximum(r["base"], r["arm"]), r["obj"])) < 1e-4


def _check_contacts():
    """Contact geometry of the kernel vs the oracle on identical states (same list order, same geom pairs)."""
    c = rc.state_synchronised.contacts
    depth, pos, cosn = np.array(c["depth"]), np.array(c["pos"]), np.array(c["cosn"])
    print(f"contacts compared: {c['n']} (steps with differing pair lists: {c['mismatched_steps']}); |ddist| p99 {np.percentile(depth, 99):.1e} "
          f"max {depth.max():.1e}; |dpos| p99 {np.percentile(pos, 99):.1e} max {pos.max():.1e}; normals within 0.5 deg: {(cosn > 0.99996).mean():.4f}, "
          f"within 20 deg: {(cosn > 0.94).mean():.4f}")
    assert c["n"] > 500
    assert np.percentile(depth, 99) < 5e-5 and np.percentile(pos, 99) < 5e-4
    # faceted hull pairs and curved rims: MPR's portal facet is round-off sensitive (in fp64 too); everything else is tight
    assert (cosn > 0.99996).mean() > 0.97 and (cosn > 0.0).mean() > 0.995


def _check_events(rel, events):
    S = rc.state_synchronised
    ro, ob, sr, so = S.rel_robot, S.rel_obj, S.same_robot, S.same_obj
    pc = lambda a: f"p50 {np.percentile(a, 50):.1e} p99 {np.percentile(a, 99):.1e} max {a.max():.1e}"
    print(f"\none-step relative qacc error over {len(rel)} env-steps, robot dofs / object dofs each on their own scale: robot {pc(ro)}; objects {pc(ob)}; "
          f"with the kernel's contact list handed to the oracle: robot {pc(sr)}; objects {pc(so)}; events {len(events)}")
    gross = [ev for ev in events if ev["eps"] != "contact-point scatter"]
    for ev in gross:
        print("   ", ev)
    clean, cr = S.clean, S.clean_robot
    print(f"   on the {len(clean)} env-steps whose contact lists agree (same pairs, normals within 0.5 deg): robot p99 {np.percentile(cr, 99):.1e} max {cr.max():.1e}")
    assert len(clean) > 0.9 * len(rel) and np.percentile(cr, 99) < rc.TYPICAL_TOL
    # objects, on their own scale, every step: on identical contacts (the dynamics) and on each side's own narrowphase
    assert np.percentile(so, 99) < rc.OBJ_TOL and np.percentile(sr, 99) < rc.TYPICAL_TOL, (pc(so), pc(sr))
    assert np.percentile(ob, 99) < rc.RAW_OBJ_TOL, pc(ob)
    assert all(ev["flags"] == 0 for ev in events)
    unexplained = [ev for ev in events if not ev["explained"]]
    assert not unexplained, unexplained
    assert len(gross) <= 0.005 * len(rel) + 2


@pytest.mark.parametrize("scene", SCENES)
def test_emul_random_ctrl_free_running(scene):
    blob, model = _blob(scene)
    be = rc.EmulBackend(blob, 6)
    r = rc.free_running(be, blob, model, 6, 10, seed=11)
    _report(f"emulator {scene}", r)
    _check_free_running(r, scene)


@pytest.mark.parametrize("scene", SCENES)
def test_emul_state_synchronised_steps(scene):
    blob, model = _blob(scene)
    be = rc.EmulBackend(blob, 4)
    rel, events = rc.state_synchronised(be, blob, model, 4, 5, seed=5, twin=_keeps_manifolds(model))   # (where the kernel keeps manifolds: against the oracle's twin of that rule)
    _check_events(rel, events)
    _check_contacts()


@pytest.mark.gpu
@pytest.mark.parametrize("scene", SCENES + ["stretch_kitchen_robocasa"])
def test_gpu_random_ctrl_free_running_1000_steps(scene):
    """16 heterogeneous envs x 1000 steps of the bench's action schedule on the HIP path, each against its own oracle -- north_star's
    criterion on every bench scene, the kitchen at Robocasa scale (satellite builds, 82 dofs) included; an env outside 1e-4 must come
    with an explained departure."""
    blob, model = _blob(scene)
    be = rc.HipBackend(scene, 16)
    r = rc.free_running(be, blob, model, 16, 20, seed=7)
    be.close()
    _report(f"HIP {scene}", r)
    _check_free_running(r, scene)


@pytest.mark.gpu
@pytest.mark.parametrize("scene", SCENES)
def test_gpu_state_synchronised_steps(scene):
    """8 envs x 400 steps with the oracle's state uploaded before every step: per-step errors and explained events."""
    blob, model = _blob(scene)
    be = rc.HipBackend(scene, 8)
    rel, events = rc.state_synchronised(be, blob, model, 8, 8, seed=3, twin=_keeps_manifolds(model))
    be.close()
    _check_events(rel, events)
    _check_contacts()
