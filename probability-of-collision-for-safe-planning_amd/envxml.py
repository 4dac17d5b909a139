"""OpenRAVE environment XML -> 2-D collision world (row N1 of SURVEY 8f).

Reads the subset of the format the reference's scenes use (pr2test2.env.xml:12-80,
pr2custom.env.xml:12-240): <KinBody> / <Body> / <Geom type="box"> with <Extents> (half sizes),
<Translation>, <RotationAxis x y z deg>; body- and kinbody-level <Translation>/<RotationAxis>
are composed with the geometry's; <offsetfrom> only names the parent body (all parents in the
reference scenes sit at the origin) and is resolved for translation.  Only rotations about z are
meaningful for the planar model; anything else raises.  KinBodies that reference an external
file (`file="data/ikeatable.kinbody.xml"`, not part of the reference tree) are skipped and
reported.  A box is kept when its z range overlaps [z_min, z_max] (default: the PR2 base lifted
to z = 0.05 up to 1.5 m), which drops the floor and the door lintel.
"""
import math
import xml.etree.ElementTree as ET

import numpy as np


def _floats(node, tag, n, default=None):
    el = node.find(tag) if node is not None else None
    if el is None or el.text is None:
        return default
    vals = [float(v) for v in el.text.split()]
    if len(vals) != n:
        raise ValueError("<%s> expects %d numbers, got %r" % (tag, n, el.text))
    return vals


def _yaw(node):
    ra = None
    if node is not None:
        el = node.find("RotationAxis")
        if el is None:
            el = node.find("rotationaxis")
        if el is not None:
            ra = [float(v) for v in el.text.split()]
    if ra is None:
        return 0.0
    ax, ay, az, deg = ra
    if abs(deg) < 1e-12:
        return 0.0
    if abs(ax) > 1e-9 or abs(ay) > 1e-9:
        raise ValueError("only rotations about z are supported (got axis %r)" % (ra[:3],))
    return math.radians(deg) * (1.0 if az >= 0 else -1.0)


def _compose(t_parent, yaw_parent, t_child, yaw_child):
    c, s = math.cos(yaw_parent), math.sin(yaw_parent)
    return ([t_parent[0] + c * t_child[0] - s * t_child[1],
             t_parent[1] + s * t_child[0] + c * t_child[1],
             t_parent[2] + t_child[2]], yaw_parent + yaw_child)


def load_env_geoms(path):
    """Every box geometry of the scene BEFORE the z filter, as the OpenRAVE adapter sees them
    (link transform x geometry transform): list of (name, R row-major 9, t 3, half extents 3).
    Feeds csrc/scene_boxes.hpp in tests/test_scene_boxes_cpp.py."""
    geoms = []
    env = load_env_xml(path, z_min=-1e300, z_max=1e300, _geoms=geoms)
    assert len(geoms) == len(env["boxes"])
    return geoms


def load_env_xml(path, z_min=0.05, z_max=1.5, footprint=(0.0, 0.0, 0.334, 0.334), _geoms=None):
    """Returns dict(footprint, boxes M x 5 [cx cy hx hy yaw], skipped=[names], robot_start)."""
    root = ET.parse(str(path)).getroot()
    boxes, skipped = [], []
    for kb in root.findall("KinBody"):
        name = kb.get("name", "?")
        if kb.get("file"):
            skipped.append("%s (external file %s)" % (name, kb.get("file")))
            continue
        kt = _floats(kb, "Translation", 3, [0.0, 0.0, 0.0])
        ky = _yaw(kb)
        body_pose = {}
        for body in kb.findall("Body"):
            bt = _floats(body, "Translation", 3, [0.0, 0.0, 0.0])
            by = _yaw(body)
            parent = body.find("offsetfrom")
            base_t, base_y = (kt, ky)
            if parent is not None and parent.text and parent.text.strip() in body_pose:
                base_t, base_y = body_pose[parent.text.strip()]
            wt, wy = _compose(base_t, base_y, bt, by)
            if body.get("name"):
                body_pose[body.get("name")] = (wt, wy)
            for geom in body.findall("Geom"):
                if geom.get("type", "").lower() != "box":
                    skipped.append("%s/%s geom type %s" % (name, body.get("name", "?"), geom.get("type")))
                    continue
                ext = _floats(geom, "Extents", 3)
                if ext is None:
                    ext = _floats(geom, "extents", 3)
                gt = _floats(geom, "Translation", 3, [0.0, 0.0, 0.0])
                (cx, cy, cz), yaw = _compose(wt, wy, gt, _yaw(geom))
                if cz + ext[2] < z_min or cz - ext[2] > z_max:
                    continue                                  # floor, lintel, ...
                boxes.append([cx, cy, ext[0], ext[1], yaw])
                if _geoms is not None:
                    c, s = math.cos(yaw), math.sin(yaw)
                    _geoms.append(("%s/%s" % (name, body.get("name", "?")), [c, -s, 0.0, s, c, 0.0, 0.0, 0.0, 1.0],
                                   [cx, cy, cz], list(ext)))
    start = None
    rob = root.find("Robot")
    if rob is not None:
        start = _floats(rob, "translation", 3) or _floats(rob, "Translation", 3)
    return dict(footprint=list(footprint), boxes=np.array(boxes, dtype=np.float64).reshape(-1, 5),
                skipped=skipped, robot_start=start)


def write_env_txt(env, path, comment=""):
    """The text format read by planio.load_env."""
    with open(path, "w") as f:
        f.write("# pocs env v1%s\n" % (" -- " + comment if comment else ""))
        f.write("footprint %.17g %.17g %.17g %.17g\n" % tuple(env["footprint"]))
        for b in np.asarray(env["boxes"]).reshape(-1, 5):
            f.write("box %.17g %.17g %.17g %.17g %.17g\n" % tuple(b))
