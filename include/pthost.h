/* pthost.h — C ABI of the host side that stays above the hot path (libpthost.so): the scene_v3.xml loader,
 * RGB -> sigmoid-polynomial spectrum upsampling and the `zout` film writer. It is the C++ stand-in for the
 * reference's Rust host (this image has no Rust toolchain): pt/src/parser/mod.rs:26-255 (parse),
 * pt/src/spectrum/mod.rs:26-50,271-302 (Spectrum::new), pt/src/camera.rs:197-237 (Film::save_image).
 * It produces exactly the plain-data structures include/ptgpu.h consumes. No GPU code in here.
 */
#ifndef PTHOST_H
#define PTHOST_H
#include "ptgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pthost_scene pthost_scene;

const char* pthost_last_error(void);

/* parser::parse(scene_collection, scene_base_path): opens <collection>/<name>/scene_v3.xml.
 * overrides: "key=value;key=value" replacing <default name=key> values (e.g. "spp=64;resx=512;integrator=guided"),
 * may be NULL. data_dir: directory holding icosphere_6.ply (NULL = <libdir>/../data). Returns NULL on error.
 * Deviation (documented in DESIGN.md): bsdfs and shapes keep XML document order where the reference iterates
 * HashMaps in per-process random order. */
pthost_scene* pthost_scene_load(const char* collection, const char* name, const char* overrides, const char* data_dir);
void pthost_scene_free(pthost_scene* s);

const ptgpu_scene_desc* pthost_scene_desc(const pthost_scene* s);
const ptgpu_camera_desc* pthost_camera_desc(const pthost_scene* s);
const ptgpu_integrator_desc* pthost_integrator_desc(const pthost_scene* s);
uint32_t pthost_spp(const pthost_scene* s);
const char* pthost_output_name(const pthost_scene* s);

/* Spectrum::new(RGB | Constant, illuminant) up to the table lookup: grey/black colours use the reference's closed
 * form; other colours are fitted directly (Jakob-Hanika Gauss-Newton in CIELAB under D65) because the
 * reference's srgb.coeffs table is missing from its tree (.MISSING_LARGE_BLOBS). Returns 0 on success. */
int pthost_rgb_to_spectrum(const float rgb[3], ptgpu_spectrum* out);

/* Film::save_image: film = UNSCALED linear RGB sums (ptgpu_render output), scale = film_scale.
 * as_splats != 0 applies the splat transfer (gamma 1/2.2, lightpath), else the tile transfer (gamma 1/2.4).
 * Writes RGBA8 PNG to <dir>/<file_name> (dir created if missing). */
int pthost_save_image(const double* film_rgb, uint32_t resx, uint32_t resy, double scale, int as_splats, const char* dir,
                      const char* file_name);
/* Film::save_image with both of the reference's buffers (either may be NULL = all zero): splat pixels are written with
 * gamma 1/2.2, tile pixels are then added with gamma 1/2.4 as u8 (wrapping, as `+=` does in a release build). */
int pthost_save_image_film(const double* tile_rgb, const double* splat_rgb, uint32_t resx, uint32_t resy, double scale, const char* dir,
                           const char* file_name);
/* Linear float dump (PFM, little endian, bottom-up rows as the format requires) for relMSE tooling. */
int pthost_save_pfm(const double* film_rgb, uint32_t resx, uint32_t resy, double scale, const char* path);

#ifdef __cplusplus
}
#endif
#endif
