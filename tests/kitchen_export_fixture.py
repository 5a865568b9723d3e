"""TEST FIXTURE -- a hand-written document in the style of a robosuite / Robocasa kitchen export (what
`env.sim.model.get_xml()` gives the reference's generator, robocasa_gen.py:196-239; Robocasa itself and its assets are not
available here).  It uses what such exports use and stretch.xml does not: <inertial>, visual geoms in group 1 (massless: mass="0") beside
collision geoms in group 0, articulated fixtures (a hinged cabinet door, a sliding drawer), capsule
(fromto) and ellipsoid geoms, marker geoms / sites, and the robosuite robot with its actuators and sensors (which the
converter drops)."""

KITCHEN_EXPORT = """<mujoco model="kitchen_export">
  <compiler angle="radian"/>
  <option timestep="0.001" integrator="Euler" cone="pyramidal"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 0.05" rgba="0.6 0.6 0.6 1"/>
    <body name="counter_main" pos="0.0 -1.10 0.45">
      <inertial pos="0 0 0" mass="80" diaginertia="8 8 8"/>
      <geom name="counter_col" type="box" size="1.2 0.3 0.45" group="0" rgba="0.55 0.4 0.3 1"/>
      <geom name="counter_vis" type="box" size="1.21 0.31 0.455" group="1" contype="0" conaffinity="0" mass="0" rgba="0.8 0.8 0.75 1"/>
      <geom name="counter_reg" type="box" size="1.0 0.25 0.01" pos="0 0 0.46" rgba="0.5 0 0 0.5" contype="0" conaffinity="0" group="1" mass="0"/>
      <site name="counter_site" pos="0 0 0.5" rgba="0.5 0 0 1"/>
      <body name="door" pos="0.6 0.31 0.0">
        <inertial pos="-0.2 0.01 0" mass="2" diaginertia="0.05 0.03 0.03"/>
        <joint name="door_hinge" type="hinge" axis="0 0 1" pos="0 0 0" range="0 1.6" damping="0.5" frictionloss="0.05"/>
        <geom name="door_col" type="box" size="0.2 0.01 0.35" pos="-0.2 0.01 0" group="0" rgba="0.5 0.35 0.25 1"/>
        <geom name="door_handle" type="capsule" size="0.008" fromto="-0.36 0.05 -0.08 -0.36 0.05 0.08" group="0" rgba="0.8 0.8 0.8 1"/>
      </body>
      <body name="drawer" pos="-0.6 0.05 0.25">
        <inertial pos="0 0 0" mass="1.5" diaginertia="0.02 0.02 0.03"/>
        <joint name="drawer_slide" type="slide" axis="0 1 0" range="0 0.3" damping="2" frictionloss="0.2"/>
        <geom name="drawer_col" type="box" size="0.25 0.26 0.08" group="0" rgba="0.5 0.35 0.25 1"/>
        <geom name="drawer_handle" type="capsule" size="0.008" fromto="-0.08 0.29 0 0.08 0.29 0" group="0" rgba="0.8 0.8 0.8 1"/>
      </body>
    </body>
    <body name="bottle_main" pos="-0.25 -0.93 0.9505">
      <freejoint name="bottle_joint0"/>
      <inertial pos="0 0 -0.005" mass="0.35" diaginertia="0.0009 0.0009 0.0003"/>
      <geom name="bottle_col" type="cylinder" size="0.03 0.05" group="0" rgba="0.2 0.5 0.8 1" friction="0.9 0.005 0.0001"/>
      <geom name="bottle_neck" type="capsule" size="0.012" fromto="0 0 0.05 0 0 0.09" group="0" rgba="0.2 0.5 0.8 1"/>
      <geom name="bottle_vis" type="cylinder" size="0.031 0.051" group="1" contype="0" conaffinity="0" mass="0" rgba="0.2 0.5 0.8 1"/>
    </body>
    <body name="lemon_main" pos="0.05 -0.93 0.9305">
      <freejoint name="lemon_joint0"/>
      <geom name="lemon_col" type="ellipsoid" size="0.045 0.036 0.03" group="0" mass="0.12" rgba="0.9 0.85 0.2 1"/>
    </body>
    <body name="spatula_main" pos="0.35 -0.95 0.9125">
      <freejoint name="spatula_joint0"/>
      <geom name="spatula_col" type="capsule" size="0.012" fromto="-0.09 0 0 0.05 0.0 0" group="0" mass="0.06" rgba="0.3 0.3 0.3 1"/>
      <geom name="spatula_blade" type="box" size="0.035 0.03 0.004" pos="0.09 0 -0.008" group="0" mass="0.04" rgba="0.3 0.3 0.3 1"/>
    </body>
    <body name="robot0_base" pos="0.0 -0.2 0" quat="1 0 0 0">
      <joint name="robot0_joint_mobile_forward" type="slide" axis="1 0 0"/>
      <geom name="robot0_g0" type="box" size="0.2 0.2 0.2" mass="10"/>
      <body name="robot0_link1" pos="0 0 0.4"><joint name="robot0_joint1" type="hinge" axis="0 0 1"/><geom name="robot0_g1" type="sphere" size="0.05" mass="1"/></body>
    </body>
  </worldbody>
  <contact><exclude body1="robot0_base" body2="robot0_link1"/><exclude body1="counter_main" body2="door"/><exclude body1="counter_main" body2="drawer"/></contact>
  <actuator><motor name="robot0_m1" joint="robot0_joint1"/></actuator>
  <sensor><jointpos name="robot0_s1" joint="robot0_joint1"/></sensor>
</mujoco>"""
