"""Rectangular material region of the layer stack.

Mirrors the reference input type `Material` (mesh_and_materials/materials.py:1-37):
same constructor arguments, attribute names and validation errors, so the
drivers and a user's own scripts can build the same objects.  `tag` / `_tag`
are attached by the mesher (reference: mesh_and_materials/mesh.py:114,126).
"""


class Material:
    """A named box ``[zmin, zmax, rmin, rmax]`` with properties and a target mesh size.

    ``properties`` carries ``{"rho_cv": rho*cv, "k": k}`` for the heat solver
    (reference: run_with_diamond.py:100-181).
    """

    def __init__(self, name, boundaries, properties=None, mesh_size=None, material_tag=None):
        if not isinstance(name, str):
            raise TypeError(f"name must be a string, got {type(name)}")
        if not hasattr(boundaries, "__len__") or len(boundaries) != 4:
            raise ValueError("boundaries must be [xmin,xmax,ymin,ymax]")
        lo_x, hi_x, lo_y, hi_y = (float(b) for b in boundaries)
        if lo_x >= hi_x or lo_y >= hi_y:
            raise ValueError(f"Invalid boundaries {boundaries}")
        if mesh_size is not None and not isinstance(mesh_size, (int, float)):
            raise TypeError(f"mesh_size must be a number, got {type(mesh_size)}")
        self.name = name
        self.boundaries = [lo_x, hi_x, lo_y, hi_y]
        self.mesh_size = None if mesh_size is None else float(mesh_size)
        self.properties = dict(properties) if properties is not None else {}
        if material_tag is not None:
            self.tag = material_tag
            self._tag = material_tag

    def contains(self, x, y):
        """True when (x, y) lies in the closed box."""
        lo_x, hi_x, lo_y, hi_y = self.boundaries
        return lo_x <= x <= hi_x and lo_y <= y <= hi_y

    def __repr__(self):
        return f"Material({self.name!r}, bounds={self.boundaries}, size={self.mesh_size})"
