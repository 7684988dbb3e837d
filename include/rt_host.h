/* rt_host.h -- host-side companion of rt_abi.h (librt_host.so, no GPU code).
 *
 * A scene loader that accepts the reference's `.scene` / `.obj` grammar
 * (parseSceneFile, raytracer.cpp:705-841; parseObj, raytracer.cpp:624-703;
 * transform builders algebra3.h:1259-1299; mat4::inverse algebra3.h:1130-1161)
 * and produces the flat rt_scene_desc that rt_upload_scene takes, plus the
 * PNG writer standing where Film::writeFile calls lodepng::encode
 * (raytracer.cpp:205-235).  The reference itself would not use this loader:
 * it would flatten its own Scene object (oracle/ref_wrapper.cpp does exactly that walk).
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_scene rth_scene;   /* owns every array its rt_scene_desc points to */

/* Parse `scene_path`.  `obj` file names are resolved against the scene file's
 * directory (the reference resolves them against the cwd, raytracer.cpp:629,
 * and is run from that directory).  max_depth < 0 selects the reference's 6
 * (raytracer.cpp:477); eps <= 0 selects main's 0.05f (raytracer.cpp:849). */
int rth_load_scene(rth_scene** out, const char* scene_path, int width, int height,
                   int max_depth, float eps);

/* Read / write the serialised "RTSD" form described in rt_abi.h. */
int rth_load_blob(rth_scene** out, const char* blob_path);
int rth_save_blob(const rth_scene* scene, const char* blob_path);

const rt_scene_desc* rth_desc(const rth_scene* scene);
void rth_free_scene(rth_scene* scene);

/* Write an RGBA8 image whose row 0 is the top scanline -- the buffer
 * rt_download fills -- as an 8-bit RGBA PNG. */
int rth_write_png(const char* path, const uint8_t* rgba, int width, int height);

/* Where scanline tile k of an N-GPU job lies in the frame (the layout of rt_download and
 * rt_bind_frame: RGBA8, row 0 = top scanline): tile k is film rows [k * tile_rows,
 * min((k + 1) * tile_rows, height)) -- film row 0 is the BOTTOM scanline (raytracer.cpp:153, 225) --
 * and belongs to rank k % n_ranks (rt_render_tiles(ctx, rank, tile_rows, n_ranks)).  Its rows are
 * contiguous in the frame: *offset is the byte offset of the tile's topmost row, *bytes its length.
 * Returns the owning rank, or RT_ERR_ARG when k is not a tile of the frame.  The end-of-frame
 * exchange sends exactly these spans to rank 0 (no counterpart in the reference, which has one
 * process and no exchange). */
int rth_tile_span(int width, int height, int tile_rows, int n_ranks, int k, size_t* offset, size_t* bytes);

/* The packed form of that exchange (one message per peer): what rank `rank` of `n_ranks` owns, as a
 * strided region of the frame.  Its full tiles (tile_rows rows each) lie at a constant stride of
 * n_ranks tiles: *n_full of them, the lowest at byte offset *first_offset, each *tile_bytes long,
 * *stride_bytes apart -- one 2-D copy (hipMemcpy2DAsync: width = tile_bytes, height = n_full,
 * source pitch = stride_bytes) packs them into a contiguous buffer in ascending frame order.  The
 * frame's last tile is clipped when the height is not a multiple of tile_rows; if this rank owns it,
 * *clip_bytes > 0 and it lies at *clip_offset (it follows the full tiles in the packed buffer).
 * Returns 0, or RT_ERR_ARG. */
int rth_packed_share(int width, int height, int tile_rows, int n_ranks, int rank, int* n_full, size_t* first_offset,
                     size_t* tile_bytes, size_t* stride_bytes, size_t* clip_offset, size_t* clip_bytes);

const char* rth_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
