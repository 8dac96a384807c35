"""Layer-stack geometry: cfg dict -> Material boxes ``[zmin, zmax, rmin, rmax]``.

Axis convention (reference run_with_diamond.py:76-96, :321-322): mesh x == z
(axial), mesh y == r (radial), the sample mid-plane is z = 0, the p-side is
negative z.  All lengths in metres.

Every number read from the YAML goes through ``float(...)``: PyYAML loads
mantissas without a dot (``5e-6``) as *strings* (SURVEY.md section 5), and the
reference drivers coerce the same way (run_with_diamond.py:60-74).
"""
from __future__ import annotations

from dataclasses import dataclass, field

from .materials import Material


def _f(cfg, mat, key):
    return float(cfg["mats"][mat][key])


def _material(cfg, name, box):
    """Material with rho_cv = rho*cv and k, as run_with_diamond.py:100-181."""
    return Material(
        name,
        boundaries=box,
        properties={"rho_cv": _f(cfg, name, "rho") * _f(cfg, name, "cv"), "k": _f(cfg, name, "k")},
        mesh_size=_f(cfg, name, "mesh"),
    )


@dataclass
class Stack:
    """Materials in the reference's list order plus the derived landmarks."""

    materials: list
    bounds: list                      # [zmin, zmax, rmin, rmax] handed to Mesh(...)
    heated_z: float                   # z of the Gaussian-heated line (p_coupler.boundaries[0])
    r_sample: float
    kind: str                         # "with_diamond" | "no_diamond"
    extra: dict = field(default_factory=dict)
    heated_z_oside: float = 0.0       # extension (two-sided heating): outer face of the o-side coupler

    def by_name(self, name):
        for m in self.materials:
            if m.name == name:
                return m
        raise KeyError(name)


def stack_with_diamond(cfg) -> Stack:
    """Nine-box diamond-anvil-cell stack (reference run_with_diamond.py:60-181).

    The diamonds span the full radius r_sample + r_gasket + r_ins_gside; the
    insulators, couplers and sample span r in [0, r_sample]; g_ins and gasket
    fill the annulus between the diamonds.
    """
    r_sample = _f(cfg, "p_sample", "r")
    r_gask = _f(cfg, "gasket", "r")
    r_gins = _f(cfg, "g_ins", "r")
    r_max = r_sample + r_gask + r_gins

    t_oins = _f(cfg, "o_ins", "z")
    t_pins = _f(cfg, "p_ins", "z")
    t_samp = _f(cfg, "p_sample", "z")
    t_coup = _f(cfg, "p_coupler", "z")
    t_diam = _f(cfg, "p_diam", "z")

    z_lo = -(t_samp / 2) - t_pins - t_coup - t_diam
    z_hi = (t_samp / 2) + t_oins + t_coup + t_diam

    # Same additions in the same order as the reference so the break points are
    # the same doubles (they feed np.isclose() in the BC location).
    p_diam = [z_lo, z_lo + t_diam, 0.0, r_max]
    o_diam = [z_hi - t_diam, z_hi, 0.0, r_max]
    p_ins = [p_diam[1], p_diam[1] + t_pins, 0.0, 0.0 + r_sample]
    o_ins = [o_diam[0] - t_oins, o_diam[0], 0.0, 0.0 + r_sample]
    p_coup = [p_ins[1], p_ins[1] + t_coup, 0.0, 0.0 + r_sample]
    o_coup = [o_ins[0] - t_coup, o_ins[0], 0.0, 0.0 + r_sample]
    sample = [p_coup[1], p_coup[1] + t_samp, 0.0, 0.0 + r_sample]
    g_ins = [p_diam[1], o_diam[0], 0.0 + r_sample, 0.0 + r_sample + r_gins]
    gasket = [p_diam[1], o_diam[0], g_ins[3], r_max]

    boxes = [
        ("p_diam", p_diam), ("p_ins", p_ins), ("p_coupler", p_coup), ("p_sample", sample),
        ("o_coupler", o_coup), ("o_ins", o_ins), ("o_diam", o_diam), ("gasket", gasket), ("g_ins", g_ins),
    ]
    mats = [_material(cfg, n, b) for n, b in boxes]
    return Stack(mats, [z_lo, z_hi, 0.0, r_max], heated_z=p_coup[0], heated_z_oside=o_coup[1], r_sample=r_sample,
                 kind="with_diamond", extra={"z_ins_pside": t_pins, "z_coupler": t_coup})


def stack_no_diamond(cfg) -> Stack:
    """Five-box stack without diamonds/gasket (reference run_no_diamond.py:62-129).

    ``bounds[3]`` = r_sample + r_ins_oside is what the reference passes to
    ``Mesh(...)`` (run_no_diamond.py:76,132-136) but only the material boxes are
    meshed, so the triangulated domain ends at max(rmax of the boxes).
    """
    r_sample = _f(cfg, "p_sample", "r")
    r_oins = _f(cfg, "o_ins", "r")
    r_coup = _f(cfg, "p_coupler", "r")
    r_pins = _f(cfg, "p_ins", "r")

    t_oins = _f(cfg, "o_ins", "z")
    t_pins = _f(cfg, "p_ins", "z")
    t_samp = _f(cfg, "p_sample", "z")
    t_coup = _f(cfg, "p_coupler", "z")

    z_lo = -(t_samp / 2) - t_pins - t_coup
    z_hi = (t_samp / 2) + t_oins + t_coup

    p_ins = [z_lo, z_lo + t_pins, 0.0, 0.0 + r_pins]
    p_coup = [p_ins[1], p_ins[1] + t_coup, 0.0, 0.0 + r_coup]
    sample = [p_coup[1], p_coup[1] + t_samp, 0.0, 0.0 + r_sample]
    o_coup = [sample[1], sample[1] + t_coup, 0.0, 0.0 + r_coup]
    o_ins = [o_coup[1], o_coup[1] + t_oins, 0.0, 0.0 + r_oins]

    boxes = [("p_ins", p_ins), ("p_coupler", p_coup), ("p_sample", sample), ("o_coupler", o_coup), ("o_ins", o_ins)]
    mats = [_material(cfg, n, b) for n, b in boxes]
    return Stack(mats, [z_lo, z_hi, 0.0, r_sample + r_oins], heated_z=p_coup[0], heated_z_oside=o_coup[1],
                 r_sample=r_sample, kind="no_diamond", extra={"z_ins_pside": t_pins, "z_coupler": t_coup})


def build_stack(cfg) -> Stack:
    """Pick the stack by the presence of ``p_diam`` (as parameter_sweep.py:92)."""
    return stack_with_diamond(cfg) if "p_diam" in cfg["mats"] else stack_no_diamond(cfg)


def watcher_points(cfg) -> dict:
    """Mid-plane of each coupler at r = 0 (reference with_diamond.py:15-37,
    no_diamond.py, parameter_sweep.py:69-120).  Values are (z, r)."""
    t_samp = _f(cfg, "p_sample", "z")
    t_pins = _f(cfg, "p_ins", "z")
    t_oins = _f(cfg, "o_ins", "z")
    t_coup = _f(cfg, "p_coupler", "z")
    t_diam = _f(cfg, "p_diam", "z") if "p_diam" in cfg["mats"] else None
    if t_diam is not None:
        z_lo = -(t_samp / 2) - t_pins - t_coup - t_diam
        z_hi = (t_samp / 2) + t_oins + t_coup + t_diam
        p_ins_end = z_lo + t_diam + t_pins
        o_ins_start = z_hi - t_diam - t_oins
    else:
        z_lo = -(t_samp / 2) - t_pins - t_coup
        z_hi = (t_samp / 2) + t_oins + t_coup
        p_ins_end = z_lo + t_pins
        o_ins_start = z_hi - t_oins
    return {"pside": (p_ins_end + t_coup / 2, 0.0), "oside": (o_ins_start - t_coup / 2, 0.0)}


def scale_mesh_sizes(cfg, factor):
    """Deep-copied cfg with every ``mats.*.mesh`` multiplied by ``factor``
    ("all mesh: scaled by one factor so n = 1.0e6")."""
    import copy

    out = copy.deepcopy(cfg)
    for m in out["mats"].values():
        m["mesh"] = float(m["mesh"]) * float(factor)
    return out
