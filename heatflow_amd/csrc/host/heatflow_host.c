/* libheatflow_host.so - host-only helpers of the mesh layer (see include/heatflow_host.h). */
#include "heatflow_host.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int hfh_version(void) { return 1; }

/* decimal digits of an int, returns the new write position */
static char* put_int(char* p, long v) {
  char tmp[24];
  int k = 0;
  if (v < 0) { *p++ = '-'; v = -v; }
  if (v == 0) tmp[k++] = '0';
  while (v > 0) { tmp[k++] = (char)('0' + v % 10); v /= 10; }
  while (k > 0) *p++ = tmp[--k];
  return p;
}

int hfh_write_msh22(const char* path, int32_t n, int32_t ne, const double* coords, const int32_t* tris,
                    const int32_t* tags, int32_t n_names, const char* const* names, const int32_t* name_tags) {
  if (!path || n < 0 || ne < 0 || (n > 0 && !coords) || (ne > 0 && (!tris || !tags))) return -EINVAL;
  FILE* f = fopen(path, "w");
  if (!f) return -errno;
  const size_t cap = 1u << 20;
  char* buf = (char*)malloc(cap + 256);
  if (!buf) { fclose(f); return -ENOMEM; }
  int rc = 0;
  fprintf(f, "$MeshFormat\n2.2 0 8\n$EndMeshFormat\n");
  if (n_names > 0 && names && name_tags) {
    fprintf(f, "$PhysicalNames\n%d\n", n_names);
    for (int32_t k = 0; k < n_names; ++k) fprintf(f, "2 %d \"%s\"\n", name_tags[k], names[k]);
    fprintf(f, "$EndPhysicalNames\n");
  }
  fprintf(f, "$Nodes\n%d\n", n);
  char* p = buf;
  for (int32_t i = 0; i < n; ++i) {
    p = put_int(p, (long)i + 1);
    p += snprintf(p, 64, " %.17g %.17g 0\n", coords[2 * (size_t)i], coords[2 * (size_t)i + 1]);
    if ((size_t)(p - buf) > cap) { if (fwrite(buf, 1, (size_t)(p - buf), f) != (size_t)(p - buf)) rc = -EIO; p = buf; }
  }
  if (p != buf && fwrite(buf, 1, (size_t)(p - buf), f) != (size_t)(p - buf)) rc = -EIO;
  fprintf(f, "$EndNodes\n$Elements\n%d\n", ne);
  p = buf;
  for (int32_t e = 0; e < ne; ++e) {
    p = put_int(p, (long)e + 1);
    memcpy(p, " 2 2 ", 5); p += 5;
    p = put_int(p, tags[e]); *p++ = ' ';
    p = put_int(p, tags[e]);
    for (int a = 0; a < 3; ++a) { *p++ = ' '; p = put_int(p, (long)tris[3 * (size_t)e + a] + 1); }
    *p++ = '\n';
    if ((size_t)(p - buf) > cap) { if (fwrite(buf, 1, (size_t)(p - buf), f) != (size_t)(p - buf)) rc = -EIO; p = buf; }
  }
  if (p != buf && fwrite(buf, 1, (size_t)(p - buf), f) != (size_t)(p - buf)) rc = -EIO;
  fprintf(f, "$EndElements\n");
  free(buf);
  if (fclose(f) != 0 && rc == 0) rc = -EIO;
  return rc;
}
