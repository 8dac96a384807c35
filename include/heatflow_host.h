/* heatflow_host.h - C ABI of libheatflow_host.so: host-only helpers of the mesh layer (plain C, no GPU).
 *
 * The reference writes and reads its meshes through gmsh's C++ core (mesh_and_materials/mesh.py:174-195
 * `gmsh.write(filename)`, run_with_diamond.py:219-245 `gmshio.read_from_msh`); the Python text writer that
 * replaced it here was the largest single cost of a stock run (0.9 of 1.7 s), so the file writer is native too.
 * All functions return 0 on success, a negative errno-style code otherwise.
 */
#ifndef HEATFLOW_HOST_H
#define HEATFLOW_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int hfh_version(void);

/* Gmsh MSH 2.2 ASCII file (replaces gmsh.write, mesh.py:174-195): nodes (x = z, y = r, z = 0, 17 significant
 * digits: exact round trip) and 3-node triangles (type 2) whose physical-group and surface tags are the
 * material tag.  names/name_tags (n_names entries, may be 0) become the $PhysicalNames section.
 * coords: n x 2 doubles, tris: ne x 3 zero-based int32, tags: ne int32. */
int hfh_write_msh22(const char* path, int32_t n, int32_t ne, const double* coords, const int32_t* tris,
                    const int32_t* tags, int32_t n_names, const char* const* names, const int32_t* name_tags);

/* Quadtree level map of the built-in mesher (replaces the gmsh size field + meshing of mesh.py:81-149; the numpy
 * statement of the same algorithm stays in heatflow_amd/mesh.py and is compared in the tests).  mat / allowed /
 * level: int8 maps over the padded base grid, row-major [nzp][nrp], nzp and nrp multiples of 2^lmax; mat < 0 =
 * outside every material.  level[c] becomes the quadtree level of the leaf holding base cell c (-1 outside): the
 * largest aligned block that is one material and no coarser than allowed, then 2:1-balanced with smooth grading.
 * Returns -EDOM if the balance does not settle. */
int hfh_quadtree_levels(int32_t nzp, int32_t nrp, int32_t lmax, const int8_t* mat, const int8_t* allowed, int8_t* level);
/* Leaves of a level map, ordered by level and row-major within a level: writes up to cap triples, returns the count. */
int64_t hfh_quadtree_leaves(int32_t nzp, int32_t nrp, int32_t lmax, const int8_t* level, int64_t cap, int64_t* i0,
                            int64_t* j0, int64_t* lev);

/* Nodes and triangles of the mesh from the quadtree's leaves (second half of Mesh.build_mesh, replaces gmsh's meshing of
 * mesh.py:81-149 together with the functions above).  i0 / j0 / lev: the leaves (hfh_quadtree_leaves); mat: the padded
 * material map; zc (nz + 1) / rc (nr + 1): the base grid's coordinates.  A leaf with a neighbour's corner on an edge midpoint
 * ("hanging node") becomes a fan around its centre, every other leaf two right triangles; nodes are numbered and triangles
 * listed in Morton order (low 16 bits of the lattice indices: nzp, nrp <= 65535, else -EINVAL and the caller's numpy path),
 * triangles counter-clockwise in (z, r), tag = material index + 1.  -EDOM: a degenerate triangle.  The result is held by
 * the handle: sizes, then fetch into caller arrays (coords n x 2, node_ij n x 2, tris nt x 3, tags nt), then free. */
typedef struct hfh_mesh hfh_mesh;
int hfh_mesh_build(int64_t nleaf, const int64_t* i0, const int64_t* j0, const int64_t* lev, int32_t nzp, int32_t nrp,
                   const int8_t* mat, int32_t nz, int32_t nr, const double* zc, const double* rc, hfh_mesh** out);
int hfh_mesh_sizes(const hfh_mesh* h, int64_t* n_nodes, int64_t* n_tris, int64_t* n_fan);
int hfh_mesh_fetch(const hfh_mesh* h, double* coords, int64_t* node_ij, int32_t* tris, int32_t* tags);
void hfh_mesh_free(hfh_mesh* h);

#ifdef __cplusplus
}
#endif
#endif
