"""TEST FIXTURE -- a generated document at the scale of a Robocasa kitchen export (what `env.sim.model.get_xml()` hands the
reference's generator, robocasa_gen.py:196-239; Robocasa, robosuite and their assets are not available here, so the document is
written by this script instead of exported).  What it has in common with the real thing, because that is what the import and
the kernels have to cope with:

  * ~45 fixture bodies welded to the world (cabinets, counters, sink, stove, hood, fridge, microwave, dishwasher, island, table,
    chairs, shelves, walls), each with SEVERAL collision geoms -- box panels and convex MESH pieces (inline <mesh vertex=...>
    assets: handles, faucet, hood, knobs, seats) -- more than 300 collision geoms in all, in robosuite's collision colour
    (rgba 0.5 0 0 1: the reference's clean-up makes them invisible, robocasa_gen.py:249-250) beside massless visual geoms in group 1;
  * 8 articulated fixture parts: cabinet doors and a fridge door (hinges), drawers (slides), stove knobs (hinges), with ranges,
    damping and friction loss;
  * 8 free objects on the counters, the island and the table (can, cereal box, bowl (mesh), mug, apple, bottle, tray, sponge);
  * marker geoms / sites in the colours the clean-up looks for, robosuite's robot body `robot0_base` with actuators and sensors
    (removed by the import), an <option> section (dropped), <contact><exclude> pairs between a fixture and its moving parts.
`kitchen_xml()` returns the document; `STATS` what it contains."""
import math

COL = 'rgba="0.5 0 0 1"'   # robosuite's collision-geom colour


def _f(v):
    return " ".join(f"{x:.6g}" for x in v)


class _Doc:
    def __init__(self):
        self.meshes, self.body, self.excl = [], [], []
        self.ncol = self.nvis = self.nmesh_geom = self.nfix = self.nart = self.nfree = 0

    def mesh(self, name, verts):
        self.meshes.append(f'<mesh name="{name}" vertex="{_f([c for v in verts for c in v])}"/>')

    def col(self, name, typ, size=None, pos=(0, 0, 0), mesh=None, extra=""):
        self.ncol += 1
        if mesh:
            self.nmesh_geom += 1
            return f'<geom name="{name}" type="mesh" mesh="{mesh}" pos="{_f(pos)}" group="0" {COL} {extra}/>'
        return f'<geom name="{name}" type="{typ}" size="{_f(size)}" pos="{_f(pos)}" group="0" {COL} {extra}/>'

    def vis(self, name, size, pos=(0, 0, 0), rgba="0.7 0.6 0.5 1"):
        self.nvis += 1
        return f'<geom name="{name}" type="box" size="{_f(size)}" pos="{_f(pos)}" group="1" contype="0" conaffinity="0" mass="0" rgba="{rgba}"/>'


def prism(n, r, h, axis=2, r2=None):
    """Vertices of an n-gon prism (or frustum: top radius r2) of half height h about `axis`."""
    out = []
    for z, rr in ((-h, r), (h, r if r2 is None else r2)):
        for k in range(n):
            a = 2 * math.pi * k / n
            p = [rr * math.cos(a), rr * math.sin(a), z]
            if axis == 0:
                p = [p[2], p[0], p[1]]
            elif axis == 1:
                p = [p[0], p[2], p[1]]
            out.append(p)
    return out


def chamfer_box(hx, hy, hz, c):
    """A box with its four long (x) edges cut: 16 vertices, convex."""
    out = []
    for sx in (-1, 1):
        for (y, z) in ((hy - c, hz), (hy, hz - c), (hy, -hz + c), (hy - c, -hz), (-hy + c, -hz), (-hy, -hz + c), (-hy, hz - c), (-hy + c, hz)):
            out.append([sx * hx, y, z])
    return out


def panels(d, name, w, dp, h, t=0.02, shelves=1, front_open=True):
    """Carcass of a cabinet of half sizes (w, dp, h) about the body origin: sides, back, bottom, top, shelves; the front stays open
    (doors / drawers close it)."""
    g = [d.col(f"{name}_left", "box", (t / 2, dp, h), (-w + t / 2, 0, 0)), d.col(f"{name}_right", "box", (t / 2, dp, h), (w - t / 2, 0, 0)),
         d.col(f"{name}_back", "box", (w, t / 2, h), (0, -dp + t / 2, 0)), d.col(f"{name}_bottom", "box", (w, dp, t / 2), (0, 0, -h + t / 2)),
         d.col(f"{name}_top", "box", (w, dp, t / 2), (0, 0, h - t / 2))]
    for k in range(shelves):
        z = -h + (k + 1) * 2 * h / (shelves + 1)
        g.append(d.col(f"{name}_shelf{k}", "box", (w - t, dp - t, t / 2), (0, 0, z)))
    if not front_open:
        g.append(d.col(f"{name}_front", "box", (w, t / 2, h), (0, dp - t / 2, 0)))
    return g


def kitchen_xml():
    d = _Doc()
    d.mesh("handle_bar", chamfer_box(0.07, 0.008, 0.008, 0.003))
    d.mesh("handle_bar_v", [[v[1], v[2], v[0]] for v in chamfer_box(0.07, 0.008, 0.008, 0.003)])
    d.mesh("knob", prism(12, 0.022, 0.012, axis=1))
    d.mesh("faucet_base", prism(10, 0.03, 0.03))
    d.mesh("faucet_neck", prism(8, 0.012, 0.12))
    d.mesh("faucet_spout", prism(8, 0.011, 0.08, axis=1))
    d.mesh("hood_lower", prism(4, 0.42, 0.06, r2=0.30))
    d.mesh("hood_upper", prism(4, 0.16, 0.20, r2=0.14))
    d.mesh("seat", prism(8, 0.17, 0.02))
    d.mesh("bowl_hull", prism(12, 0.045, 0.03, r2=0.08))
    d.mesh("toaster_hull", chamfer_box(0.13, 0.08, 0.09, 0.02))
    d.mesh("kettle_hull", prism(10, 0.075, 0.09, r2=0.05))
    d.mesh("pot_hull", prism(12, 0.10, 0.06))
    B = d.body

    def fixture(name, pos, geoms, children=""):
        d.nfix += 1
        B.append(f'<body name="{name}" pos="{_f(pos)}">' + "".join(geoms) + children + "</body>")

    def door(parent, name, pos, half, hinge_x, handle_side, rng=(0, 1.6), axis="0 0 1", handle="handle_bar_v"):
        """A door panel of half sizes `half` whose hinge runs along z at x = hinge_x of the panel."""
        d.nart += 1
        d.excl.append((parent, name))
        hx = -0.8 * half[0] * handle_side
        return (f'<body name="{name}" pos="{_f((pos[0] + hinge_x, pos[1], pos[2]))}">'
                f'<inertial pos="{_f((-hinge_x, 0, 0))}" mass="2.5" diaginertia="0.06 0.04 0.03"/>'
                f'<joint name="{name}_hinge" type="hinge" axis="{axis}" range="{_f(rng)}" damping="0.8" frictionloss="0.08" armature="0.001"/>'
                + d.col(f"{name}_panel", "box", half, (-hinge_x, 0, 0)) + d.col(f"{name}_handle", None, pos=(-hinge_x + hx, half[1] + 0.03, 0), mesh=handle)
                + d.vis(f"{name}_vis", (half[0], half[1] * 0.9, half[2]), (-hinge_x, 0, 0), "0.62 0.45 0.3 1") + "</body>")

    def drawer(parent, name, pos, half):
        d.nart += 1
        d.excl.append((parent, name))
        return (f'<body name="{name}" pos="{_f(pos)}"><inertial pos="0 0 0" mass="1.8" diaginertia="0.03 0.03 0.04"/>'
                f'<joint name="{name}_slide" type="slide" axis="0 1 0" range="0 0.32" damping="4" frictionloss="0.4"/>'
                + d.col(f"{name}_front", "box", (half[0], 0.01, half[2]), (0, half[1], 0)) + d.col(f"{name}_floor", "box", (half[0] - 0.02, half[1], 0.006), (0, 0, -half[2] + 0.01))
                + d.col(f"{name}_sl", "box", (0.006, half[1], half[2] - 0.01), (-half[0] + 0.02, 0, 0)) + d.col(f"{name}_sr", "box", (0.006, half[1], half[2] - 0.01), (half[0] - 0.02, 0, 0))
                + d.col(f"{name}_handle", None, pos=(0, half[1] + 0.035, 0), mesh="handle_bar") + d.vis(f"{name}_vis", (half[0], 0.011, half[2]), (0, half[1], 0), "0.62 0.45 0.3 1") + "</body>")

    # ---- room
    fixture("walls", (0, 0, 1.25), [d.col("wall_n", "box", (3.0, 0.05, 1.25), (0, -2.05, 0)), d.col("wall_s", "box", (3.0, 0.05, 1.25), (0, 3.05, 0)),
                                    d.col("wall_w", "box", (0.05, 2.6, 1.25), (-3.05, 0.5, 0)), d.col("wall_e", "box", (0.05, 2.6, 1.25), (3.05, 0.5, 0)),
                                    d.vis("wall_n_vis", (3.0, 0.04, 1.25), (0, -2.06, 0), "0.9 0.9 0.85 1")])
    # backsplash tiles along the north wall (behind the counter run)
    fixture("backsplash", (0, -1.97, 1.15), [d.col(f"tile{k}", "box", (0.2, 0.012, 0.22), (-2.2 + 0.4 * k, 0, 0)) for k in range(12)])
    # ---- counter run along the north wall: 6 base cabinets (0.6 wide), y from -2.0 (wall) to -1.4 (front), top at 0.88
    art_doors = {0: 1, 3: -1}      # cabinets with an articulated door (hinge side)
    art_drawers = {1: 2}           # cabinet with two articulated drawers
    for i in range(6):
        x = -2.1 + 0.6 * i + 0.3
        name = f"cab{i}"
        geoms = panels(d, name, 0.3, 0.3, 0.40, shelves=1) + [d.col(f"{name}_toe", "box", (0.3, 0.26, 0.04), (0, -0.02, -0.44)),
                                                              d.vis(f"{name}_vis", (0.295, 0.295, 0.40), (0, 0, 0), "0.6 0.45 0.3 1")]
        ch = ""
        if i in art_doors:
            s = art_doors[i]
            ch = door(name, f"{name}_door", (0, 0.31, 0), (0.29, 0.01, 0.39), 0.29 * s, s, rng=(0, 1.7) if s < 0 else (-1.7, 0))
        elif i in art_drawers:
            ch = drawer(name, f"{name}_drawer0", (0, 0.04, 0.2), (0.27, 0.26, 0.09)) + drawer(name, f"{name}_drawer1", (0, 0.04, -0.12), (0.27, 0.26, 0.16))
        else:
            geoms.append(d.col(f"{name}_fdoor", "box", (0.29, 0.01, 0.39), (0, 0.31, 0)))
            geoms.append(d.col(f"{name}_fhandle", None, pos=(0.2, 0.35, 0.2), mesh="handle_bar_v"))
        fixture(name, (x, -1.7, 0.48), geoms, ch)
    fixture("countertop_n", (-0.3, -1.69, 0.90), [d.col("ct_n_a", "box", (0.95, 0.33, 0.02), (-0.85, 0, 0)), d.col("ct_n_b", "box", (0.85, 0.33, 0.02), (0.95, 0, 0)),
                                                  d.col("ct_n_lip", "box", (1.8, 0.01, 0.03), (0.0, -0.32, 0.05)),
                                                  d.vis("ct_n_vis", (1.8, 0.33, 0.02), (0, 0, 0), "0.85 0.85 0.8 1"),
                                                  '<geom name="ct_n_region" type="box" size="0.8 0.25 0.005" pos="0.9 0 0.03" rgba="0.5 0 0 0.5" contype="0" conaffinity="0" group="1" mass="0"/>',
                                                  '<site name="ct_n_site" pos="0 0 0.05" size="0.01" rgba="0.5 0 0 1"/>'])
    # sink between the two slabs, with a faucet of three convex pieces
    fixture("sink", (-0.25, -1.69, 0.80), [d.col("sink_floor", "box", (0.24, 0.2, 0.008), (0, 0, -0.07)), d.col("sink_l", "box", (0.008, 0.2, 0.08), (-0.24, 0, 0)),
                                           d.col("sink_r", "box", (0.008, 0.2, 0.08), (0.24, 0, 0)), d.col("sink_f", "box", (0.24, 0.008, 0.08), (0, 0.2, 0)),
                                           d.col("sink_b", "box", (0.24, 0.008, 0.08), (0, -0.2, 0)), d.col("faucet_base", None, pos=(0, -0.26, 0.15), mesh="faucet_base"),
                                           d.col("faucet_neck", None, pos=(0, -0.26, 0.3), mesh="faucet_neck"), d.col("faucet_spout", None, pos=(0, -0.18, 0.41), mesh="faucet_spout"),
                                           d.vis("sink_vis", (0.24, 0.2, 0.01), (0, 0, -0.07), "0.75 0.75 0.8 1")])
    # stove with four articulated knobs, hood above it
    knobs = ""
    fixed_knobs = []
    for k in range(4):
        if k >= 2:   # (two of the four knobs turn; the others are part of the stove)
            fixed_knobs.append(d.col(f"stove_knob{k}_g", None, pos=(-0.3 + 0.2 * k, 0.325, 0.32), mesh="knob"))
            continue
        d.nart += 1
        d.excl.append(("stove", f"stove_knob{k}"))
        knobs += (f'<body name="stove_knob{k}" pos="{_f((-0.3 + 0.2 * k, 0.325, 0.32))}"><inertial pos="0 0 0" mass="0.05" diaginertia="2e-5 2e-5 2e-5"/>'
                  f'<joint name="stove_knob{k}_hinge" type="hinge" axis="0 1 0" range="-0.1 2.4" damping="0.02" frictionloss="0.01" armature="1e-4"/>'
                  + d.col(f"stove_knob{k}_g", None, mesh="knob") + "</body>")
    fixture("stove", (1.9, -1.7, 0.46), panels(d, "stove", 0.38, 0.31, 0.44, shelves=0, front_open=False) + [d.col(f"burner{k}", "cylinder", (0.09, 0.008), ((-0.18, 0.18)[k % 2], (-0.13, 0.13)[k // 2], 0.448)) for k in range(4)]
            + fixed_knobs + [d.vis("stove_vis", (0.38, 0.31, 0.44), (0, 0, 0), "0.3 0.3 0.32 1")], knobs)
    fixture("hood", (1.9, -1.78, 1.62), [d.col("hood_lower", None, mesh="hood_lower"), d.col("hood_upper", None, pos=(0, -0.05, 0.26), mesh="hood_upper"), d.vis("hood_vis", (0.3, 0.22, 0.05), (0, 0, 0), "0.6 0.6 0.62 1")])
    # wall cabinets above the counter (one with an articulated door)
    for i in range(5):
        name = f"wcab{i}"
        geoms = panels(d, name, 0.3, 0.17, 0.32, shelves=2) + [d.vis(f"{name}_vis", (0.295, 0.165, 0.32), (0, 0, 0), "0.6 0.45 0.3 1")]
        ch = ""
        if i == 2:
            ch = door(name, f"{name}_door", (0, 0.18, 0), (0.29, 0.01, 0.31), -0.29, -1, rng=(0, 1.7))
        else:
            geoms.append(d.col(f"{name}_fdoor", "box", (0.29, 0.01, 0.31), (0, 0.18, 0)))
            geoms.append(d.col(f"{name}_fhandle", None, pos=(-0.2, 0.22, -0.2), mesh="handle_bar_v"))
        fixture(name, (-2.1 + 0.6 * i + 0.3, -1.83, 1.78), geoms, ch)
    # fridge (carcass + one articulated door, one fixed freezer door), microwave on a shelf, dishwasher
    fixture("fridge", (-2.55, 0.2, 0.95), panels(d, "fridge", 0.38, 0.36, 0.93, shelves=3) + [d.col("fridge_freezer", "box", (0.37, 0.012, 0.3), (0, 0.372, 0.62)),
            d.col("fridge_fhandle", None, pos=(0.3, 0.41, 0.45), mesh="handle_bar_v"), d.vis("fridge_vis", (0.38, 0.36, 0.93), (0, 0, 0), "0.88 0.88 0.9 1")],
            door("fridge", "fridge_door", (0, 0.372, -0.31), (0.37, 0.012, 0.6), -0.37, -1, rng=(0, 2.0)))
    fixture("microwave", (0.9, -1.8, 1.28), panels(d, "microwave", 0.26, 0.19, 0.15, shelves=0) + [d.col("microwave_fdoor", "box", (0.2, 0.008, 0.14), (-0.05, 0.198, 0)),
            d.col("microwave_panel", "box", (0.05, 0.008, 0.14), (0.21, 0.198, 0)), d.vis("microwave_vis", (0.26, 0.19, 0.15), (0, 0, 0), "0.2 0.2 0.22 1")])
    fixture("dishwasher", (2.55, -0.6, 0.43), panels(d, "dishwasher", 0.3, 0.3, 0.42, shelves=2, front_open=False) + [d.col("dishwasher_handle", None, pos=(0, 0.33, 0.3), mesh="handle_bar"),
            d.vis("dishwasher_vis", (0.3, 0.3, 0.42), (0, 0, 0), "0.7 0.7 0.72 1")])
    # island with a slab, table with four legs, four chairs, two stools with mesh seats, a shelf unit
    fixture("island", (1.3, 0.9, 0.44), panels(d, "island", 0.5, 0.35, 0.42, shelves=1, front_open=False) + [d.col("island_top", "box", (0.6, 0.45, 0.02), (0, 0, 0.44)),
            d.vis("island_vis", (0.5, 0.35, 0.42), (0, 0, 0), "0.5 0.5 0.55 1")])
    fixture("table", (-0.8, 1.9, 0.0), [d.col("table_top", "box", (0.7, 0.45, 0.02), (0, 0, 0.74))] + [d.col(f"table_leg{k}", "box", (0.025, 0.025, 0.36), ((-0.62, 0.62)[k % 2], (-0.38, 0.38)[k // 2], 0.36)) for k in range(4)]
            + [d.col("table_apron", "box", (0.6, 0.01, 0.04), (0, 0.36, 0.68)), d.vis("table_vis", (0.7, 0.45, 0.02), (0, 0, 0.74), "0.6 0.45 0.3 1")])
    for c, (x, y) in enumerate(((-1.3, 1.3), (-0.3, 1.3), (-1.3, 2.55), (-0.3, 2.55))):
        fixture(f"chair{c}", (x, y, 0), [d.col(f"chair{c}_seat", "box", (0.2, 0.2, 0.015), (0, 0, 0.45)), d.col(f"chair{c}_back", "box", (0.2, 0.015, 0.22), (0, 0.19 if y > 2 else -0.19, 0.7))]
                + [d.col(f"chair{c}_leg{k}", "box", (0.015, 0.015, 0.22), ((-0.17, 0.17)[k % 2], (-0.17, 0.17)[k // 2], 0.22)) for k in range(4)] + [d.vis(f"chair{c}_vis", (0.2, 0.2, 0.015), (0, 0, 0.45), "0.45 0.3 0.2 1")])
    for c, (x, y) in enumerate(((0.9, 1.65), (1.7, 1.65))):
        fixture(f"stool{c}", (x, y, 0), [d.col(f"stool{c}_seat", None, pos=(0, 0, 0.62), mesh="seat")] + [d.col(f"stool{c}_leg{k}", "box", (0.012, 0.012, 0.3), ((-0.11, 0.11)[k % 2], (-0.11, 0.11)[k // 2], 0.3)) for k in range(4)]
                + [d.col(f"stool{c}_ring", "box", (0.12, 0.12, 0.008), (0, 0, 0.2)), d.vis(f"stool{c}_vis", (0.15, 0.15, 0.02), (0, 0, 0.62), "0.3 0.3 0.3 1")])
    fixture("shelfunit", (-2.8, 1.9, 0.9), [d.col("shelf_l", "box", (0.15, 0.01, 0.9), (0, -0.5, 0)), d.col("shelf_r", "box", (0.15, 0.01, 0.9), (0, 0.5, 0))]
            + [d.col(f"shelf_b{k}", "box", (0.15, 0.5, 0.01), (0, 0, -0.85 + 0.42 * k)) for k in range(5)] + [d.vis("shelf_vis", (0.15, 0.5, 0.9), (0, 0, 0), "0.6 0.45 0.3 1")])
    # east counter run with two plain cabinets, small appliances on the counters
    for i in range(2):
        name = f"ecab{i}"
        fixture(name, (2.7, 0.35 + 0.62 * i, 0.48), panels(d, name, 0.3, 0.3, 0.40, shelves=1, front_open=False) + [d.col(f"{name}_handle", None, pos=(-0.33, 0, 0.2), mesh="handle_bar_v"),
                d.vis(f"{name}_vis", (0.295, 0.295, 0.40), (0, 0, 0), "0.6 0.45 0.3 1")])
    fixture("countertop_e", (2.68, 0.66, 0.90), [d.col("ct_e", "box", (0.33, 0.64, 0.02)), d.vis("ct_e_vis", (0.33, 0.64, 0.02), (0, 0, 0), "0.85 0.85 0.8 1")])
    fixture("toaster", (-1.6, -1.78, 1.01), [d.col("toaster_g", None, mesh="toaster_hull"), d.col("toaster_lever", "box", (0.01, 0.015, 0.02), (0.14, 0, 0.02)), d.vis("toaster_vis", (0.13, 0.08, 0.09), (0, 0, 0), "0.8 0.8 0.8 1")])
    fixture("kettle", (2.7, 0.3, 1.01), [d.col("kettle_g", None, mesh="kettle_hull"), d.col("kettle_handle", "box", (0.01, 0.05, 0.05), (0, -0.1, 0.02)), d.vis("kettle_vis", (0.07, 0.07, 0.09), (0, 0, 0), "0.75 0.75 0.78 1")])
    fixture("pot", (1.72, -1.57, 0.97), [d.col("pot_g", None, mesh="pot_hull"), d.col("pot_h0", "box", (0.03, 0.01, 0.008), (0.12, 0, 0.03)), d.col("pot_h1", "box", (0.03, 0.01, 0.008), (-0.12, 0, 0.03)), d.vis("pot_vis", (0.1, 0.1, 0.06), (0, 0, 0), "0.4 0.4 0.42 1")])
    fixture("rack", (-1.0, -1.9, 1.05), [d.col(f"rack_bar{k}", "box", (0.2, 0.006, 0.006), (0, 0, -0.08 + 0.04 * k)) for k in range(5)] + [d.col("rack_l", "box", (0.006, 0.02, 0.1), (-0.2, 0, 0)), d.col("rack_r", "box", (0.006, 0.02, 0.1), (0.2, 0, 0))])
    # odds and ends: bin, utensil holder, fruit basket, paper-towel holder, wall shelf, window sill, lamp, doormat
    fixture("bin", (2.5, 2.3, 0.3), [d.col("bin_floor", "box", (0.18, 0.18, 0.01), (0, 0, -0.29))] + [d.col(f"bin_w{k}", "box", ((0.18, 0.01)[k % 2], (0.01, 0.18)[k % 2], 0.3), ((0, 0.18, 0, -0.18)[k], (0.18, 0, -0.18, 0)[k], 0)) for k in range(4)])
    fixture("utensils", (2.72, 0.95, 0.99), [d.col("utensils_cup", None, mesh="faucet_base"), d.col("utensils_a", "box", (0.004, 0.004, 0.09), (0.01, 0, 0.1)), d.col("utensils_b", "box", (0.004, 0.004, 0.08), (-0.01, 0.01, 0.09))])
    fixture("basket", (1.0, 1.05, 0.93), [d.col("basket_floor", "box", (0.12, 0.09, 0.005), (0, 0, -0.025))] + [d.col(f"basket_w{k}", "box", ((0.12, 0.005)[k % 2], (0.005, 0.09)[k % 2], 0.03), ((0, 0.12, 0, -0.12)[k], (0.09, 0, -0.09, 0)[k], 0)) for k in range(4)])
    fixture("towels", (-0.7, -1.9, 1.08), [d.col("towels_roll", "cylinder", (0.055, 0.12)), d.col("towels_base", "cylinder", (0.07, 0.006), (0, 0, -0.126)), d.col("towels_rod", "box", (0.005, 0.005, 0.14), (0, 0, 0.0))])
    fixture("wallshelf", (-3.0, -0.9, 1.5), [d.col("wallshelf_board", "box", (0.12, 0.45, 0.012)), d.col("wallshelf_br0", "box", (0.1, 0.01, 0.06), (0, -0.35, -0.07)), d.col("wallshelf_br1", "box", (0.1, 0.01, 0.06), (0, 0.35, -0.07)),
                                              d.col("wallshelf_jar0", "cylinder", (0.04, 0.06), (0.02, -0.2, 0.072)), d.col("wallshelf_jar1", "cylinder", (0.04, 0.06), (0.02, 0.0, 0.072)), d.col("wallshelf_jar2", "cylinder", (0.04, 0.06), (0.02, 0.2, 0.072))])
    fixture("windowsill", (0.6, 2.95, 1.0), [d.col("sill_board", "box", (0.6, 0.08, 0.015)), d.col("sill_pot0", None, pos=(-0.3, 0, 0.075), mesh="kettle_hull"), d.col("sill_pot1", None, pos=(0.3, 0, 0.075), mesh="kettle_hull")])
    fixture("lamp", (1.3, 0.9, 2.1), [d.col("lamp_shade", None, mesh="hood_upper"), d.col("lamp_cord", "box", (0.004, 0.004, 0.2), (0, 0, 0.4))])
    fixture("doormat", (0.0, 2.7, 0.004), [d.col(f"mat_e{k}", "box", ((0.4, 0.01)[k % 2], (0.01, 0.25)[k % 2], 0.004), ((0, 0.4, 0, -0.4)[k], (0.25, 0, -0.25, 0)[k], 0)) for k in range(4)])
    # ---- free objects (robosuite style: <name>_main body, <name>_joint0 free joint, collision geoms in group 0, visual in group 1)
    def obj(name, pos, geoms, mass_note=""):
        d.nfree += 1
        B.append(f'<body name="{name}_main" pos="{_f(pos)}"><freejoint name="{name}_joint0"/>' + "".join(geoms) + "</body>")
    fr = 'friction="0.95 0.3 0.1" solref="0.02 1" solimp="0.95 0.99 0.001"'
    obj("can", (0.55, -1.55, 0.975), [d.col("can_g", "cylinder", (0.033, 0.055), extra=f'mass="0.35" {fr}'), d.vis("can_vis", (0.03, 0.03, 0.055), rgba="0.8 0.1 0.1 1")])
    obj("cereal", (1.28, -1.6, 1.0305), [d.col("cereal_g", "box", (0.1, 0.035, 0.11), extra=f'mass="0.4" {fr}'), d.vis("cereal_vis", (0.1, 0.035, 0.11), rgba="0.9 0.7 0.2 1")])
    obj("bowl", (-1.15, -1.55, 0.9505), [d.col("bowl_g", None, mesh="bowl_hull", extra=f'mass="0.3" {fr}'), d.vis("bowl_vis", (0.07, 0.07, 0.03), rgba="0.9 0.9 0.95 1")])
    obj("mug", (0.2, -1.55, 0.9655), [d.col("mug_g", "cylinder", (0.04, 0.045), extra=f'mass="0.25" {fr}'), d.col("mug_h", "box", (0.012, 0.006, 0.025), (0.052, 0, 0), extra='mass="0.02"'), d.vis("mug_vis", (0.04, 0.04, 0.045), rgba="0.2 0.4 0.8 1")])
    obj("apple", (1.2, 0.8, 0.9405), [d.col("apple_g", "sphere", (0.04,), extra=f'mass="0.15" condim="4" friction="0.9 0.01 0.001"'), d.vis("apple_vis", (0.035, 0.035, 0.035), rgba="0.8 0.15 0.1 1")])
    obj("bottle", (1.55, 1.0, 1.0005), [d.col("bottle_g", "cylinder", (0.035, 0.1), extra=f'mass="0.5" {fr}'), d.col("bottle_neck", "capsule", (0.014, 0.03), (0, 0, 0.125), extra='mass="0.03"'), d.vis("bottle_vis", (0.035, 0.035, 0.1), rgba="0.1 0.5 0.3 1")])
    # (a tray, not a plate: a 22 cm x 2 cm cylinder lying flat never comes to rest under the multiccd manifold -- its rim contacts hop,
    # 4e-5 m per step in the fp64 oracle too -- which would only measure that artefact in every env)
    obj("tray", (-0.8, 1.8, 0.7685), [d.col("tray_g", "box", (0.12, 0.09, 0.008), extra=f'mass="0.3" {fr}'), d.vis("tray_vis", (0.12, 0.09, 0.008), rgba="0.95 0.95 0.95 1")])
    obj("sponge", (-1.2, 1.95, 0.7805), [d.col("sponge_g", "box", (0.05, 0.035, 0.02), extra=f'mass="0.05" {fr}'), d.vis("sponge_vis", (0.05, 0.035, 0.02), rgba="0.9 0.8 0.2 1")])
    # ---- robosuite's robot (removed by the import; its pose becomes Stretch's spawn pose)
    B.append('<body name="robot0_base" pos="0.3 -0.8 0" quat="1 0 0 0"><joint name="robot0_joint_mobile_forward" type="slide" axis="1 0 0"/>'
             '<geom name="robot0_g0" type="box" size="0.2 0.2 0.2" mass="10"/><body name="robot0_link1" pos="0 0 0.4"><joint name="robot0_joint1" type="hinge" axis="0 0 1"/>'
             '<geom name="robot0_g1" type="sphere" size="0.05" mass="1"/></body></body>')
    excl = "".join(f'<exclude body1="{a}" body2="{b}"/>' for a, b in d.excl) + '<exclude body1="robot0_base" body2="robot0_link1"/>'
    xml = ('<mujoco model="kitchen_robocasa_scale"><compiler angle="radian"/><option timestep="0.001" integrator="Euler" cone="pyramidal"/>'
           "<asset>" + "".join(d.meshes) + "</asset><worldbody>"
           '<geom name="floor" type="plane" size="0 0 0.05" rgba="0.6 0.6 0.6 1"/>' + "".join(B) + "</worldbody>"
           "<contact>" + excl + "</contact>"
           '<actuator><motor name="robot0_m1" joint="robot0_joint1"/></actuator><sensor><jointpos name="robot0_s1" joint="robot0_joint1"/></sensor></mujoco>')
    stats = dict(fixture_bodies=d.nfix, collision_geoms=d.ncol, mesh_collision_geoms=d.nmesh_geom, articulated=d.nart, free_objects=d.nfree, visual_geoms=d.nvis)
    return xml, stats


if __name__ == "__main__":
    x, st = kitchen_xml()
    print(st, len(x))
