/* rt_loaders.h — the data formats on either side of the render path, native and threaded.
 *
 * In the reference these are host-side, serial stages around Camera::render:
 *   parse_obj            src/obj.rs:8-41      (third-party parser: crate `wavefront` 0.2, not in the reference tree;
 *                                              its documented behaviour is restated, pinned by the tests of obj.rs:43-205)
 *   Canvas::from_ppm     src/canvas.rs:86-107 (third-party decoder: crate `image` 0.24, PNM P3 / P6)
 *   Canvas::to_ppm       src/canvas.rs:57-84  (third-party encoder: `image` PnmEncoder, ASCII samples, maxval 255)
 * With the frame at 2.5 ms and the scene set-up under a second they are what a run of the reference's CLI on this
 * library would spend its time in: an 870 000-triangle OBJ file is 60 MB of text, the ASCII PPM of a 4K frame 100 MB.
 * They stay on the host (text parsing and formatting; the 8-bit frame comes from rt_render_u8), in librt_host.so,
 * and feed the world-builder API of rt_builder.h.
 *
 * Errors: functions return a negative code; rtl_last_error() gives the message of the calling thread's last failure.
 */
#ifndef RT_LOADERS_H
#define RT_LOADERS_H

#include <stddef.h>
#include <stdint.h>

#include "rt_builder.h"

#ifdef __cplusplus
extern "C" {
#endif

const char* rtl_last_error(void);

/* parse_obj (src/obj.rs:8-41) into the world: every OBJ group of `text` becomes a child group of `group` holding the
 * group's triangles (polygons fan-triangulated around their first vertex, obj.rs:82-123; a triangle is smooth only if
 * all three corners carry a normal, obj.rs:24-33), all with `material` (NULL = Material::default()).  Groups come in
 * order of first appearance; faces before any `g` line form a default group.  `v`, `vn`, `f`, `g` are read; every
 * other statement is skipped.  threads <= 0: one per hardware thread, at most 16.
 * Returns the number of child groups added (0 for a file without faces), negative on a malformed file. */
int rtl_group_add_obj(RtbWorld* world, int group, const RtbMaterial* material, const char* text, size_t len, int threads);

/* The parse alone (tests, tools): the tables and per-group index arrays that rtl_group_add_obj hands to
 * rth_group_add_mesh.  positions / normals: 3 floats each; vidx / nidx: 3 per triangle (nidx -1 = no normal), the
 * triangles of all groups back to back in group order; group g owns triangles [group_first[g], group_first[g+1]).
 * Free with rtl_obj_free. */
typedef struct {
  uint32_t n_positions, n_normals, n_triangles, n_groups;
  const float* positions;
  const float* normals;
  const int32_t* vidx;
  const int32_t* nidx;
  const uint32_t* group_first; /* n_groups + 1 entries */
  const char* const* group_names;
  void* owner;
} RtlObj;
int rtl_obj_parse(const char* text, size_t len, int threads, RtlObj* out);
void rtl_obj_free(RtlObj*);

/* Canvas::from_ppm (src/canvas.rs:86-107): a P3 (ASCII) or P6 (binary) PPM with maxval <= 255 as f32 RGB,
 * sample / maxval in f32 (the reference decodes to 8-bit samples first, then divides).  Comments (# ...) are allowed
 * wherever the format allows them.  Writes width*height*3 floats to a buffer of the library; free with rtl_free.
 * maxval > 255 is refused: what the reference's decoder does with 16-bit samples before the division is not pinned
 * by any of its tests. */
int rtl_ppm_decode(const uint8_t* data, size_t len, uint32_t* width, uint32_t* height, float** rgb);
/* ... and straight into the world as a UvImage (src/patterns/uv_pattern.rs:113-117): returns the uv pattern handle. */
int rtl_uv_image_ppm(RtbWorld* world, const uint8_t* data, size_t len);

/* Canvas::to_ppm (src/canvas.rs:57-84) from the quantised frame (what rt_render_u8 returns): "P3\n<w> <h>\n255\n",
 * then one line per row of space-separated samples.  The exact whitespace of the reference's encoder is not pinned by
 * its tests (canvas.rs:144-156 only checks that the output is UTF-8); any PPM reader sees the same image.
 * rtl_ppm_encode formats into a buffer of the library (free with rtl_free); rtl_ppm_write streams it to a file.
 * Rows are formatted on `threads` threads (<= 0: one per hardware thread, at most 16). */
int rtl_ppm_encode(const uint8_t* rgb_u8, uint32_t width, uint32_t height, int threads, char** text, size_t* len);
int rtl_ppm_write(const char* path, const uint8_t* rgb_u8, uint32_t width, uint32_t height, int threads);

void rtl_free(void*);

#ifdef __cplusplus
}
#endif
#endif /* RT_LOADERS_H */
