"""Static-body fusion and kernel scheduling tables for the HIP path.

MuJoCo keeps welded (joint-less) bodies as separate tree nodes; the rigid-body dynamics are unchanged if
each weld group is merged into one body with the composite inertia.  The HIP kernels walk the tree level
by level with one lane per body, so fewer / shallower bodies is a direct latency win (Stretch: 38 -> 20
bodies).  Quantities MuJoCo attaches to the *original* bodies are carried over explicitly:
gravity-compensation mass + application point (`body_gcmass`, `body_gcipos`) and the per-geom
`geom_invweight0` used by contact regularisation.  tests/test_model.py checks with the fp64 oracle that the
fused model reproduces the unfused trajectories.

Also emits the `k_*` tables (levels, children, dof ancestor lists, L'DL entry schedule, dof masks, row
lists) that the kernels in csrc/ index with lane ids.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from .mjcf_compiler import JNT_FREE, GEOM_PLANE, quat2mat, quat_mul, quat_norm, mat2quat, _set_const


def fuse_static_bodies(m: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    nb = len(m["body_parentid"])
    par = m["body_parentid"]
    keep = [0] + [b for b in range(1, nb) if m["body_jntnum"][b] > 0]
    newid = {b: i for i, b in enumerate(keep)}
    anchor = np.zeros(nb, int)
    rpos = np.zeros((nb, 3))
    rquat = np.tile([1.0, 0, 0, 0], (nb, 1))
    for b in range(1, nb):
        if b in newid:
            anchor[b] = b
        else:
            p = par[b]
            anchor[b] = anchor[p]
            rpos[b] = rpos[p] + quat2mat(rquat[p]) @ m["body_pos"][b]
            rquat[b] = quat_norm(quat_mul(rquat[p], m["body_quat"][b]))
    f = {k: v.copy() for k, v in m.items()}
    nk = len(keep)
    bp = np.zeros(nk, np.int32)
    bpos = np.zeros((nk, 3)); bquat = np.tile([1.0, 0, 0, 0], (nk, 1))
    mass = np.zeros(nk); ipos = np.zeros((nk, 3)); iquat = np.tile([1.0, 0, 0, 0], (nk, 1)); inertia = np.zeros((nk, 3))
    gcm = np.zeros(nk); gcp = np.zeros((nk, 3)); gravcomp = np.zeros(nk)
    for i, b in enumerate(keep):
        if b == 0:
            continue
        p = par[b]
        bp[i] = newid[anchor[p]]
        bpos[i] = rpos[p] + quat2mat(rquat[p]) @ m["body_pos"][b]
        bquat[i] = quat_norm(quat_mul(rquat[p], m["body_quat"][b]))
        members = [x for x in range(1, nb) if anchor[x] == b]
        M = sum(m["body_mass"][x] for x in members)
        com = sum(m["body_mass"][x] * (rpos[x] + quat2mat(rquat[x]) @ m["body_ipos"][x]) for x in members) / M
        I = np.zeros((3, 3))
        for x in members:
            if m["body_mass"][x] == 0:
                continue
            R = quat2mat(rquat[x]) @ quat2mat(m["body_iquat"][x])
            c = rpos[x] + quat2mat(rquat[x]) @ m["body_ipos"][x] - com
            I += R @ np.diag(m["body_inertia"][x]) @ R.T + m["body_mass"][x] * (c @ c * np.eye(3) - np.outer(c, c))
        w, V = np.linalg.eigh(I)
        order = np.argsort(-w)
        w, V = w[order], V[:, order]
        if np.linalg.det(V) < 0:
            V[:, 2] = -V[:, 2]
        mass[i], ipos[i], iquat[i], inertia[i] = M, com, mat2quat(V), w
        g = sum(m["body_gcmass"][x] for x in members)
        gcm[i] = g
        if g != 0:
            gcp[i] = sum(m["body_gcmass"][x] * (rpos[x] + quat2mat(rquat[x]) @ m["body_gcipos"][x]) for x in members) / g
        gravcomp[i] = g / M
    f["body_parentid"] = bp
    f["body_pos"], f["body_quat"] = bpos, bquat
    f["body_mass"], f["body_ipos"], f["body_iquat"], f["body_inertia"] = mass, ipos, iquat, inertia
    f["body_gcmass"], f["body_gcipos"], f["body_gravcomp"] = gcm, gcp, gravcomp
    for k in ("body_jntadr", "body_jntnum", "body_dofadr", "body_dofnum"):
        f[k] = m[k][keep].copy()
    f["body_weldid"] = np.arange(nk, dtype=np.int32)
    root = np.zeros(nk, np.int32)
    for i in range(1, nk):
        root[i] = i if bp[i] == 0 else root[bp[i]]
    f["body_rootid"] = root
    remap = np.array([newid[anchor[b]] for b in range(nb)], np.int32)
    f["jnt_bodyid"] = remap[m["jnt_bodyid"]]
    f["dof_bodyid"] = remap[m["dof_bodyid"]]

    def rebase(prefix, has_quat=True):
        old = m[prefix + "_bodyid"]
        pos = m[prefix + "_pos"].copy()
        quat = m[prefix + "_quat"].copy()
        for i, b in enumerate(old):
            pos[i] = rpos[b] + quat2mat(rquat[b]) @ m[prefix + "_pos"][i]
            quat[i] = quat_norm(quat_mul(rquat[b], m[prefix + "_quat"][i]))
        f[prefix + "_bodyid"] = remap[old]
        f[prefix + "_pos"], f[prefix + "_quat"] = pos, quat

    # every ORIGINAL body (the reference's link names, names_json["body"]) as a rigid offset on its fused body
    f["link_fused"] = remap.copy(); f["link_relpos"] = np.array(rpos, float); f["link_relquat"] = np.array(rquat, float)
    f["geom_origbody"] = m["geom_bodyid"].copy(); f["site_origbody"] = m["site_bodyid"].copy()
    rebase("geom"); rebase("site"); rebase("cam")
    d = f["dims"].copy()
    d[3] = nk
    f["dims"] = d
    giw = m["geom_invweight0"].copy()
    _set_const(f)
    f["geom_invweight0"] = giw
    f["fused_from"] = np.array(keep, np.int32)
    return f


def _static_grid_tables(f, sp, slot, cell=0.25, maxdim=(48, 48, 16)):
    """Static collision geoms (world body) of the pairs `sp`, their uniform grid, and the (moving geom slot, static geom) -> pair
    look-up.  k_sgeom: geom ids; k_sg_cell [nsg][6]: cell range lo / hi per axis; k_grid: dims (3), then origin / cell size in
    k_grid_f; k_grid_adr [ncell + 1] / k_grid_list: static-geom indices per cell; k_spair [ncgeom][nsg]: index into k_statpair
    (the pair ids, table order) or -1 where MuJoCo's filters drop the pair."""
    gb = f["geom_bodyid"]
    sg = sorted({int(g) for p in sp for g in (f["pair_geom1"][p], f["pair_geom2"][p]) if gb[g] == 0})
    sidx = {g: i for i, g in enumerate(sg)}
    nsg, ncg = len(sg), len(slot)
    if nsg > 512:   # collision_static packs (static geom | cache slot << 9) into 16 bits (csrc/smj_sat.h, SatMem::cand)
        raise ValueError(f"{nsg} static collision geoms in the world body: the satellite builds' static broadphase holds at most 512 "
                         "(merge fixture pieces, or turn collision off for decorative geoms)")
    f["k_sgeom"] = np.array(sg + [0], np.int32); f["k_nsgeom"] = np.array([nsg], np.int32)
    f["k_statpair"] = np.array(sp + [0], np.int32); f["k_nstatpair"] = np.array([len(sp)], np.int32)
    tab = np.full((max(ncg, 1), max(nsg, 1)), -1, np.int32)
    for i, p in enumerate(sp):
        g1, g2 = int(f["pair_geom1"][p]), int(f["pair_geom2"][p])
        s_, d_ = (g1, g2) if gb[g1] == 0 else (g2, g1)
        tab[slot[d_], sidx[s_]] = i
    f["k_spair"] = tab
    margin = float(max([f["pair_margin"][p] for p in sp] + [0.0]))
    lo = np.zeros((max(nsg, 1), 3)); hi = np.zeros((max(nsg, 1), 3))
    for i, g in enumerate(sg):   # world AABB of the geom's oriented box (the world body's frame is the world frame)
        R = quat2mat(f["geom_quat"][g])
        c = f["geom_pos"][g] + R @ f["geom_aabb"][g][:3]
        h = np.abs(R) @ f["geom_aabb"][g][3:] + margin
        lo[i], hi[i] = c - h, c + h
    org = lo[:nsg].min(0) - 1e-6 if nsg else np.zeros(3)
    ext = (hi[:nsg].max(0) - org) if nsg else np.ones(3)
    h = cell
    dims = np.maximum(1, np.ceil(ext / h).astype(int))
    while np.any(dims > np.array(maxdim)):   # (a large scene: coarser cells rather than more of them)
        h *= 1.5
        dims = np.maximum(1, np.ceil(ext / h).astype(int))
    cells = [[] for _ in range(int(np.prod(dims)))]
    rng = np.zeros((max(nsg, 1), 6), np.int32)
    for i in range(nsg):
        a = np.clip(np.floor((lo[i] - org) / h).astype(int), 0, dims - 1)
        b = np.clip(np.floor((hi[i] - org) / h).astype(int), 0, dims - 1)
        rng[i] = list(a) + list(b)
        for z in range(a[2], b[2] + 1):
            for y in range(a[1], b[1] + 1):
                for x in range(a[0], b[0] + 1):
                    cells[(z * dims[1] + y) * dims[0] + x].append(i)
    adr = np.zeros(len(cells) + 1, np.int32)
    adr[1:] = np.cumsum([len(c) for c in cells])
    f["k_grid"] = np.array(list(dims), np.int32)
    f["k_grid_f"] = np.array(list(org) + [h, margin])
    f["k_grid_adr"] = adr
    f["k_grid_list"] = np.array([i for c in cells for i in c] + [0], np.int32)
    f["k_sg_cell"] = rng


def kernel_tables(f: Dict[str, np.ndarray], nsat: int = 0, static_grid: bool = False) -> Dict[str, np.ndarray]:
    """Lane-indexed scheduling tables for csrc/smj_kernels.hip (added in place, `k_` prefix).

    nsat > 0: the last `nsat` bodies are SATELLITES (find_satellites) -- the tree tables below then describe the MAIN part only
    (bodies / dofs / joints before the first satellite: lane = body and lane = dof stages of the kernels); the satellites get
    one record each (`k_sat_i`, `k_sat_f`).  Row tables (friction-loss dofs, limited joints), geoms and pairs cover everything."""
    nb_all, nv_all = len(f["body_parentid"]), len(f["dof_bodyid"])
    nb = nb_all - nsat
    nv = int(f["body_dofadr"][nb]) if nsat else nv_all
    if nb > 64 or nv > 64:
        raise ValueError("kernels map bodies / dofs to the 64 lanes of one wavefront")
    if nsat:
        _satellite_tables(f, nb, nv, nsat)
    par = np.asarray(f["body_parentid"][:nb])
    level = np.zeros(nb, np.int32)
    for b in range(1, nb):
        level[b] = level[par[b]] + 1
    f["k_body_level"] = level
    f["k_nlevel"] = np.array([level.max() + 1], np.int32)
    # pointer-jumping schedule of the kinematics stage: jump[0] = parent, jump[r+1][b] = jump[r][jump[r][b]]; after
    # ceil(log2(depth)) rounds every body has been composed up to the world
    nround = max(1, int(np.ceil(np.log2(max(int(level.max()), 1)))))
    jump = [np.asarray(par, np.int32).copy()]
    jump[0][0] = 0
    for _ in range(nround - 1):
        jump.append(jump[-1][jump[-1]])
    assert np.all(jump[-1][jump[-1]] == 0)
    f["k_body_jump"] = np.stack(jump).astype(np.int32); f["k_njump"] = np.array([nround], np.int32)
    child_adr, child_num, child_list = [], [], []
    for b in range(nb):
        ch = [c for c in range(1, nb) if par[c] == b]
        child_adr.append(len(child_list)); child_num.append(len(ch)); child_list += ch
    f["k_child_adr"] = np.array(child_adr, np.int32); f["k_child_num"] = np.array(child_num, np.int32)
    f["k_child_list"] = np.array(child_list + [0], np.int32)
    # dof ancestors (proper), nearest first; dof masks per body
    dpar = f["dof_parentid"]
    anc_adr, anc_num, anc = [], [], []
    for i in range(nv):
        chain = []
        j = dpar[i]
        while j >= 0:
            chain.append(j); j = dpar[j]
        anc_adr.append(len(anc)); anc_num.append(len(chain)); anc += chain
    f["k_dof_anc_adr"] = np.array(anc_adr, np.int32); f["k_dof_anc_num"] = np.array(anc_num, np.int32)
    f["k_dof_anc"] = np.array(anc + [0], np.int32)
    mask = np.zeros(nb, np.uint64)
    for b in range(1, nb):
        x = b
        while x > 0 and f["body_dofnum"][x] == 0:
            x = par[x]
        if x == 0:
            continue
        i = f["body_dofadr"][x] + f["body_dofnum"][x] - 1
        while i >= 0:
            mask[b] |= np.uint64(1) << np.uint64(i); i = dpar[i]
    f["k_body_dofmask_lo"] = (mask & np.uint64(0xFFFFFFFF)).astype(np.int64).astype(np.uint32).view(np.int32)
    f["k_body_dofmask_hi"] = (mask >> np.uint64(32)).astype(np.int64).astype(np.uint32).view(np.int32)
    # sparse lower-triangular entries (i >= j, j ancestor-or-self of i), padded to a multiple of 64
    ei, ej = [], []
    for i in range(nv):
        j = i
        while j >= 0:
            ei.append(i); ej.append(j); j = dpar[j]
    pad = (-len(ei)) % 64
    f["k_ldl_i"] = np.array(ei + [-1] * pad, np.int32); f["k_ldl_j"] = np.array(ej + [0] * pad, np.int32)
    f["k_fric_dof"] = np.array([k for k in range(nv_all) if f["dof_frictionloss"][k] > 0] + [0], np.int32)
    f["k_nfric"] = np.array([int((f["dof_frictionloss"] > 0).sum())], np.int32)
    lim = [j for j in range(len(f["jnt_type"])) if f["jnt_limited"][j] and f["jnt_type"][j] != JNT_FREE]
    f["k_limit_jnt"] = np.array(lim + [0], np.int32); f["k_nlimit"] = np.array([len(lim)], np.int32)
    # geoms: local rotation matrices and bounding centre in the (fused) body frame
    ng = len(f["geom_type"])
    gmat = np.zeros((ng, 9)); gcen = np.zeros((ng, 3))
    for g in range(ng):
        R = quat2mat(f["geom_quat"][g])
        gmat[g] = R.reshape(9)
        gcen[g] = f["geom_pos"][g] + R @ f["geom_center"][g]
    f["k_geom_mat"] = gmat; f["k_geom_bcenter"] = gcen
    # plane pairs first-class list (geom1 is the plane)
    pp = [p for p in range(len(f["pair_geom1"])) if f["geom_type"][f["pair_geom1"][p]] == GEOM_PLANE]
    pp_set = set(pp)
    f["k_planepair"] = np.array(pp + [0], np.int32); f["k_nplanepair"] = np.array([len(pp)], np.int32)
    # convex (non-plane) pairs: collision-geom slots for the per-step world OBB cache, pair list in table order.
    # static_grid: the geoms of the WORLD body (a kitchen's fixtures: hundreds) stay out of the cache and of the pair list the
    # kernels scan -- their world frames never change -- and are found through a uniform grid instead (k_grid_*, k_sg*): the
    # broadphase of a moving geom against them visits the cells its bounding box overlaps (csrc: StepKernel::collision_static).
    gb = f["geom_bodyid"]
    nonplane = [p for p in range(len(f["pair_geom1"])) if p not in pp_set]
    is_static_pair = lambda p: static_grid and (gb[f["pair_geom1"][p]] == 0 or gb[f["pair_geom2"][p]] == 0)
    sp = [p for p in nonplane if is_static_pair(p)]
    cp = [p for p in nonplane if not is_static_pair(p)]
    cg = sorted({int(g) for p in nonplane for g in (f["pair_geom1"][p], f["pair_geom2"][p]) if not (static_grid and gb[g] == 0)})
    if len(cg) > 128:
        raise ValueError("more than 128 moving geoms take part in non-plane collision pairs")
    if len(f["geom_hullnum"]) and (int(np.max(f["geom_hullnum"])) >= 4096 or int(np.max(f["geom_hulladr"])) >= 65536):
        raise ValueError("convex hulls: the kernel packs vertex count (< 4096) and hull address (< 65536) into one word")
    slot = {g: i for i, g in enumerate(cg)}
    f["k_cgeom"] = np.array(cg + [0], np.int32); f["k_ncgeom"] = np.array([len(cg)], np.int32)
    f["k_convpair"] = np.array(cp + [0], np.int32); f["k_nconvpair"] = np.array([len(cp)], np.int32)
    f["k_convpair_s1"] = np.array([slot[int(f["pair_geom1"][p])] for p in cp] + [0], np.int32)
    f["k_convpair_s2"] = np.array([slot[int(f["pair_geom2"][p])] for p in cp] + [0], np.int32)
    f["k_convpair_ss"] = (f["k_convpair_s1"] | (f["k_convpair_s2"] << 8)).astype(np.int32)
    f["k_convpair_rsum"] = np.array([f["geom_rbound"][f["pair_geom1"][p]] + f["geom_rbound"][f["pair_geom2"][p]] + f["pair_margin"][p]
                                     for p in cp] + [-1.0])    # padding entry: negative radius = never in range
    hv = np.asarray(f["hull_vert"], float).reshape(-1, 3)
    f["k_hull_vert4"] = np.concatenate([hv, np.zeros((len(hv), 1))], axis=1) if len(hv) else np.zeros((1, 4))   # one 16-byte load per vertex
    f["k_cgeom_half"] = np.array([f["geom_aabb"][g][3:] for g in cg] + [[0, 0, 0]], float)
    f["k_cgeom_lcen"] = np.array([f["geom_aabb"][g][:3] for g in cg] + [[0, 0, 0]], float)
    _static_grid_tables(f, sp, slot)
    # sites: local matrices
    ns = len(f["site_bodyid"])
    smat = np.zeros((ns, 9))
    for s in range(ns):
        smat[s] = quat2mat(f["site_quat"][s]).reshape(9)
    f["k_site_mat"] = smat
    # body-local inertia about the body origin as a 10-vector (same layout as cinert), in body axes
    cin = np.zeros((nb, 10))
    for b in range(1, nb):
        R = quat2mat(f["body_iquat"][b])
        I = R @ np.diag(f["body_inertia"][b]) @ R.T
        cin[b, :6] = [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]
        cin[b, 6:9] = f["body_ipos"][b]
        cin[b, 9] = f["body_mass"][b]
    f["k_body_inertia_local"] = cin
    # subtree sizes (bodies are in depth-first order, so a subtree is a contiguous id range)
    size = np.ones(nb, np.int32)
    for b in range(nb - 1, 0, -1):
        size[par[b]] += size[b]
    for b in range(1, nb):
        assert all(_is_desc(par, x, b) for x in range(b, b + size[b])), "bodies must be in depth-first order"
    f["k_body_subtreesize"] = size
    f["k_maxsubtree"] = np.array([int(max([f["k_body_subtreesize"][b] for b in range(1, nb)] + [1]))], np.int32)
    roots = [b for b in range(1, nb) if par[b] == 0 and f["body_rootid"][b] == b and f["body_subtreemass"][b] > 0]   # (main trees)
    f["k_root_list"] = np.array(roots + [0], np.int32); f["k_nroot"] = np.array([len(roots)], np.int32)
    gcb = [b for b in range(1, nb) if f["body_gcmass"][b] != 0]
    f["k_gc_body"] = np.array(gcb + [0], np.int32); f["k_ngc"] = np.array([len(gcb)], np.int32)
    # dofs whose velocity enters cdof_dot of dof d ([MJ] mj_comVel): all proper ancestors, except that the three
    # rotational dofs of a free joint all see only that joint's translational dofs
    vm = np.zeros(nv, np.uint64)
    qadr = np.full(nv, -1, np.int32)
    for i in range(nv):
        j = f["dof_jntid"][i]
        first = f["jnt_dofadr"][j]
        a = dpar[i]
        while a >= 0:
            if not (f["jnt_type"][j] == JNT_FREE and f["dof_jntid"][a] == j and a - first >= 3):
                vm[i] |= np.uint64(1) << np.uint64(a)
            a = dpar[a]
        if f["jnt_type"][j] != JNT_FREE:
            qadr[i] = f["jnt_qposadr"][j]
    f["k_dof_velmask_lo"] = (vm & np.uint64(0xFFFFFFFF)).astype(np.int64).astype(np.uint32).view(np.int32)
    f["k_dof_velmask_hi"] = (vm >> np.uint64(32)).astype(np.int64).astype(np.uint32).view(np.int32)
    f["k_dof_qposadr"] = qadr
    # static actuator moments (joint and fixed-tendon transmissions have configuration-independent moments)
    nu = len(f["actuator_trntype"])
    for a in range(nu):   # (satellites carry no actuators: find_satellites)
        js = [f["actuator_trnid"][a]] if f["actuator_trntype"][a] == 0 else \
            [f["wrap_objid"][w] for w in range(f["tendon_adr"][f["actuator_trnid"][a]], f["tendon_adr"][f["actuator_trnid"][a]] + f["tendon_num"][f["actuator_trnid"][a]])]
        assert all(f["jnt_dofadr"][j] < nv for j in js), "actuator on a satellite joint"
    mom = np.zeros((max(nu, 1), nv))
    for a in range(nu):
        gear = f["actuator_gear"][a]
        if f["actuator_trntype"][a] == 0:
            mom[a, f["jnt_dofadr"][f["actuator_trnid"][a]]] = gear
        else:
            t = f["actuator_trnid"][a]
            for w in range(f["tendon_adr"][t], f["tendon_adr"][t] + f["tendon_num"][t]):
                mom[a, f["jnt_dofadr"][f["wrap_objid"][w]]] += gear * f["wrap_prm"][w]
    f["k_act_moment"] = mom
    f["k_nldl"] = np.array([int((f["k_ldl_i"] >= 0).sum())], np.int32)
    # sparse views of the static moments: per actuator up to 4 (dof, moment) pairs, per dof up to 2 (actuator, moment)
    adof = np.full((max(nu, 1), 4), -1, np.int32); amom = np.zeros((max(nu, 1), 4))
    dact = np.full((nv, 2), -1, np.int32); dmom = np.zeros((nv, 2))
    for a in range(nu):
        nz = np.nonzero(mom[a])[0]
        if len(nz) > 4:
            raise ValueError("actuator transmission touches more than 4 dofs")
        adof[a, :len(nz)] = nz; amom[a, :len(nz)] = mom[a, nz]
    for k in range(nv):
        nz = np.nonzero(mom[:nu, k])[0] if nu else []
        if len(nz) > 2:
            raise ValueError("more than 2 actuators on one dof")
        dact[k, :len(nz)] = nz; dmom[k, :len(nz)] = mom[nz, k]
    f["k_act_dof"], f["k_act_mom"], f["k_dof_act"], f["k_dof_actmom"] = adof, amom, dact, dmom
    if any(f["body_jntnum"][b] > 2 for b in range(1, nb)):
        raise ValueError("kernels support at most 2 joints per body (or one free joint)")
    # implicitfast: d(qfrc)/d(qvel) on the mass-matrix pattern  ([MJ] mjd_smooth_vel, flg_bias=0).  Per pattern entry:
    # dof damping (diagonal), the static part sum_a bv_a m_ai m_aj over actuators that can never be force-clamped, and
    # for entries touched by ONE force-limited actuator its index and coefficient (skipped while its force is clamped)
    ne_ = len(f["k_ldl_i"])
    damp = np.zeros(ne_); dco = np.zeros(ne_); lact = np.full(ne_, -1, np.int32); lco = np.zeros(ne_)
    for e in range(ne_):
        i, j = int(f["k_ldl_i"][e]), int(f["k_ldl_j"][e])
        if i < 0:
            continue
        if i == j:
            damp[e] = f["dof_damping"][i]
        for a in range(nu):
            if f["actuator_biastype"][a] != 1:
                continue
            c = f["actuator_biasprm"][a][2] * mom[a, i] * mom[a, j]
            if c == 0:
                continue
            if f["actuator_forcelimited"][a]:
                if lact[e] >= 0:
                    raise ValueError("two force-limited actuators share a mass-matrix entry: not supported")
                lact[e], lco[e] = a, c
            else:
                dco[e] += c
    f["k_ldl_damp"], f["k_ldl_dcoef"], f["k_ldl_lact"], f["k_ldl_lcoef"] = damp, dco, lact, lco
    # geoms the lidar rays are tested against this round: visible (alpha != 0) planes and primitives
    gob = f.get("geom_origbody", f["geom_bodyid"])
    rg = [g for g in range(ng) if f["geom_type"][g] != 7 and f["geom_rgba"][g][3] != 0]
    f["k_ray_geom"] = np.array(rg + [0], np.int32); f["k_nraygeom"] = np.array([len(rg)], np.int32)
    f["k_ray_geom_origbody"] = np.array([gob[g] for g in rg] + [0], np.int32)
    # depth cameras: the geoms a MuJoCo camera draws ([MJ] geom groups 0-2, alpha > 0) and the camera frames
    grp = f.get("geom_group", np.zeros(ng, np.int32))
    rmid = f.get("geom_rmeshid", np.full(ng, -1, np.int32))
    vis = [g for g in range(ng) if grp[g] <= 2 and f["geom_rgba"][g][3] != 0 and (f["geom_type"][g] != 7 or rmid[g] >= 0)]
    f["k_rgeom"] = np.array(vis + [0], np.int32); f["k_nrgeom"] = np.array([len(vis)], np.int32)
    # lidar: geoms tested at run time = everything not welded to the laser (those are in sensor_lidar_static) with alpha > 0
    ls = f.get("sensor_lidar_site", [])
    laser_body = int(f["site_bodyid"][ls[0]]) if len(ls) else -1
    lg = [g for g in range(ng) if laser_body >= 0 and f["geom_bodyid"][g] != laser_body and f["geom_rgba"][g][3] != 0
          and (f["geom_type"][g] != 7 or rmid[g] >= 0)]
    f["k_lgeom"] = np.array(lg + [0], np.int32); f["k_nlgeom"] = np.array([len(lg)], np.int32)
    ncam = len(f.get("cam_bodyid", []))
    f["k_cam_mat"] = np.array([quat2mat(f["cam_quat"][c]).reshape(9) for c in range(ncam)] + [np.eye(3).reshape(9)])
    f["k_site_origbody"] = np.asarray(f.get("site_origbody", f["site_bodyid"]), np.int32)
    if len(f["k_site_origbody"]) == 0:
        f["k_site_origbody"] = np.zeros(1, np.int32)
    return f


# ----------------------------------------------------------------------------- satellites
# A kitchen is the robot plus many SMALL mechanisms that only meet it through contacts: free objects (6 dofs each) and the
# doors / drawers / knobs of fixtures (one hinge or slide each, on a body welded to the world).  Their mass matrix blocks are
# independent of the robot's and of each other's, so the kernels keep them out of the dense 32-column robot problem: the main
# tree runs lane = body / lane = dof as before, every satellite is ONE lane with a record of its own (kinematics, inertia,
# passive forces in closed form), and the constraint solver eliminates the satellite blocks from the Newton system (Schur
# complement on the rows that couple them to the robot) -- csrc/smj_sat.h.
SAT_MAX = 32
SAT_I = dict(body=0, jtype=1, qadr=2, dadr=3, ndof=4, jnt=5, stride=8)
SAT_F = dict(pos=0, quat=3, jpos=7, jaxis=10, q0=13, inl=14, arm=24, damp=30, stiff=36, spring=37, gcmass=38, gcipos=39, stride=44)


def find_satellites(f: Dict[str, np.ndarray], limit: int = SAT_MAX) -> int:
    """Number of trailing bodies of the fused model that can run as satellites: children of the world without children of their
    own, exactly one joint (free, hinge or slide), no actuator / equality / tendon on it, no sensor site or camera on the body.
    Only a SUFFIX of the body list qualifies (the main tree's bodies, joints, dofs and qpos entries then form a prefix, which is
    what the lane tables index); a free object declared before the robot simply stays in the main part."""
    nb = len(f["body_parentid"])
    par = f["body_parentid"]
    parents = set(int(p) for p in par[1:])
    busy = set()
    for a in range(len(f["actuator_trntype"])):
        t = int(f["actuator_trnid"][a])
        if f["actuator_trntype"][a] == 0:
            busy.add(t)
        else:
            busy.update(int(f["wrap_objid"][w]) for w in range(f["tendon_adr"][t], f["tendon_adr"][t] + f["tendon_num"][t]))
    for w in range(len(f["wrap_objid"])):
        busy.add(int(f["wrap_objid"][w]))
    for e in range(len(f["eq_obj1id"])):
        busy.add(int(f["eq_obj1id"][e]))
        if f["eq_obj2id"][e] >= 0:
            busy.add(int(f["eq_obj2id"][e]))
    sens = set(int(f["site_bodyid"][s]) for s in list(f.get("sensor_lidar_site", [])))
    if len(f.get("sensor_imu_site", [])) and int(f["sensor_imu_site"][0]) >= 0:
        sens.add(int(f["site_bodyid"][int(f["sensor_imu_site"][0])]))
    sens.update(int(b) for b in f.get("cam_bodyid", []))
    n = 0
    for b in range(nb - 1, 0, -1):
        j = int(f["body_jntadr"][b])
        ok = (par[b] == 0 and b not in parents and f["body_jntnum"][b] == 1 and int(f["jnt_type"][j]) in (JNT_FREE, 2, 3)
              and j not in busy and b not in sens and f["body_mass"][b] > 0)
        if not ok or n >= limit:
            break
        n += 1
    return n if n < nb - 1 else max(0, nb - 2)   # (keep at least one body in the main part)


def _satellite_tables(f, nb, nv, nsat):
    """k_sat_i / k_sat_f: one record per satellite (body nb + s), in body order; k_main_dims = the main part's nq, nv, nbody, njnt."""
    si = np.zeros((nsat, SAT_I["stride"]), np.int32)
    sf = np.zeros((nsat, SAT_F["stride"]))
    nq = int(f["jnt_qposadr"][int(f["body_jntadr"][nb])])
    njnt = int(f["body_jntadr"][nb])
    for s in range(nsat):
        b = nb + s
        j = int(f["body_jntadr"][b]); jt = int(f["jnt_type"][j]); qa = int(f["jnt_qposadr"][j]); da = int(f["jnt_dofadr"][j])
        nd = 6 if jt == JNT_FREE else 1
        si[s, :6] = [b, jt, qa, da, nd, j]
        r = sf[s]
        r[0:3] = f["body_pos"][b]; r[3:7] = f["body_quat"][b]; r[7:10] = f["jnt_pos"][j]; r[10:13] = f["jnt_axis"][j]
        r[13] = f["qpos0"][qa] if jt != JNT_FREE else 0.0
        R = quat2mat(f["body_iquat"][b])
        I = R @ np.diag(f["body_inertia"][b]) @ R.T
        r[14:20] = [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]; r[20:23] = f["body_ipos"][b]; r[23] = f["body_mass"][b]
        r[24:24 + nd] = f["dof_armature"][da:da + nd]; r[30:30 + nd] = f["dof_damping"][da:da + nd]
        if jt != JNT_FREE:
            r[36] = f["jnt_stiffness"][j]; r[37] = f["qpos_spring"][qa]
        r[38] = f["body_gcmass"][b]; r[39:42] = f["body_gcipos"][b]
    f["k_sat_i"], f["k_sat_f"] = si, sf
    f["k_main_dims"] = np.array([nq, nv, nb, njnt], np.int32)


def _is_desc(par, x, b):
    while x > b:
        x = par[x]
    return x == b


def prepare_for_kernels(m: Dict[str, np.ndarray], capacity: str = "auto", satellites: bool = False) -> Dict[str, np.ndarray]:
    """Fused model + kernel tables.  capacity: "auto" lets smj_create pick the step-kernel variant by the model's size (standard:
    32 dofs / 80 constraint rows / 16 contacts; big: 64 / 160 / 48); "big" asks for the big variant regardless -- for
    contact-rich scenes (fixtures all around the robot) whose steps would keep escalating out of the standard one.
    satellites: free objects and single-joint fixture parts declared after the robot run as satellites (find_satellites) -- the
    blob then addresses the satellite builds of the step kernel (the main part must fit 32 dofs / 32 bodies); "auto": only when the
    model does not fit the dense builds (what StretchBatchSimulator does for a scene given as an .xml path)."""
    f = fuse_static_bodies(m)
    if satellites == "auto":
        # the dense builds hold 64 dofs, 32 fused bodies and 128 geoms in convex pairs (smj_model.h); a scene beyond that -- a kitchen
        # with its fixtures and objects -- goes to the satellite builds when its extra bodies qualify as satellites
        pg = set(int(g) for k in ("pair_geom1", "pair_geom2") for g in f[k] if f["geom_type"][int(g)] != 0)
        satellites = (len(f["dof_bodyid"]) > 64 or len(f["body_parentid"]) > 32 or len(pg) > 128) and find_satellites(f) > 0
    nsat = find_satellites(f) if satellites else 0
    # the static-geometry tables belong to the satellite builds: a model asked for with satellites=True in which nothing qualifies as a
    # satellite runs on a dense build, whose convex-pair scan must keep the pairs against world-body geoms (collision_static is compiled
    # for NSAT > 0 only)
    f = kernel_tables(f, nsat, static_grid=nsat > 0)
    f["k_nsat"] = np.array([nsat], np.int32)
    f["k_capacity_hint"] = np.array([1 if capacity == "big" else 0], np.int32)
    return f
