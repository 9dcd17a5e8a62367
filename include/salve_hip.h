/*
 * salve_hip.h -- C ABI of libsalve_hip.so, the MI355X (gfx950) implementation of SALVe's hot path:
 * the per-hypothesis BEV texture-map rasteriser and the early-fusion ResNet verifier.
 *
 * The reference (zillow/salve @ 2024_10_08) is pure Python and has NO FFI for this path; the boundary it
 * exposes is a set of Python call sites.  Each entry point below names the reference interface it stands
 * behind (paths relative to the reference tree).  The Python facade in salve_amd/ binds these symbols with
 * ctypes and reproduces the reference's signatures, None-returns and exceptions.
 *
 * Conventions
 *   - Every pointer marked "device" is a raw HIP device pointer owned by the caller (PyTorch on the host
 *     side); the library never allocates per call and never frees caller memory.
 *   - All work is enqueued on the given hipStream_t (passed as void*; NULL = the null stream) and is
 *     asynchronous with respect to the host.
 *   - Return value: 0 = OK, negative = salve_status_t; salve_last_error() gives a thread-local message.
 *     No C++ exception crosses this boundary.
 *   - One host thread + one process per GPU is the supported model; handles are not thread-safe.
 */
#ifndef SALVE_HIP_H
#define SALVE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    SALVE_OK = 0,
    SALVE_ERR_BAD_ARG = -1,
    SALVE_ERR_UNSUPPORTED = -2,
    SALVE_ERR_HIP = -3,
    SALVE_ERR_WORKSPACE = -4
} salve_status_t;

#define SALVE_HIP_ABI_VERSION 6  /* 3: device status word (densify, resnet_forward), in-window counts from salve_bev_scatter;
                                     4: salve_bev_tile_pairs; 5: panorama index (salve_bev_pano_index_*), the scatter stage writes the
                                     sparse image into out_bev (no key image in memory, salve_bev_workspace_init is gone), salve_resnet_create
                                     takes its kernel selection as `flags` -- the library reads no environment variable;
                                     6: SALVE_RESNET_CHAIN_STORE_ALL / _NO_TRANSPOSED_TILES / _NO_NEXT_FUSE, out_flags bit 4 (renders densified in the
                                     given order), salve_bev_densify_tiles, a launch of >= 1025 renders keeps its dispatch order in the workspace's key image; unknown
                                     `flags` / `out_flags` bits are refused with SALVE_ERR_BAD_ARG (ABI 5 ignored them) */

/* Device status word: an optional device int32 the caller zeroes once and passes to the launches below.  Kernels OR bits
 * into it when something went wrong that an int return value cannot report (the launch is asynchronous); the caller
 * reads it back at a point where it synchronises anyway.  0 = every launch since the last reset was sound. */
#define SALVE_STATUS_WALK_FAILED 1 /* bev_densify: a Delaunay star did not close -- that render's image is incomplete */
#define SALVE_STATUS_FP16_RANGE 2  /* resnet_forward: an activation exceeded the fp16 range and was saturated (no released
                                      checkpoint does this; a network without normalisation can) */
#define SALVE_STATUS_BAD_HYPOTHESIS 4 /* bev scatter stage: a salve_bev_hyp_t row names a panorama outside [0, n_panos) or a surface
                                         other than 0 / 1 -- that render is an empty image */

#define SALVE_STATUS_LAYOUT_THICKNESS 8 /* layout_rasterise: a segment of thickness >= 19 pixels (OpenCV draws its end caps as 20- / 72-gons,
                                          which are not implemented) -- that segment is not drawn */

/* Library / ABI version (SALVE_HIP_ABI_VERSION). */
int salve_hip_version(void);
/* Message of the last failing call on this thread ("" if none). */
const char* salve_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * BEV rasteriser.
 * Stands behind salve/utils/bev_rendering_utils.py: get_xyzrgb_from_depth (:347-414), the pose
 * application of render_bev_pair (:443-451), render_bev_image (:254-328) and, underneath it,
 * zorder_utils.choose_elevated_repeated_vals (salve/utils/zorder_utils.py:10-83),
 * interpolation_utils.interp_dense_grid_from_sparse / remove_hallucinated_content
 * (salve/utils/interpolation_utils.py:21-54, 74-122).
 * ------------------------------------------------------------------------------------------------ */

/* The constants the reference hard-codes, gathered in one struct. */
typedef struct {
    int32_t pano_h, pano_w;     /* working pano resolution: 512 x 1024 (bev_rendering_utils.py:373-374) */
    int32_t crop_rows;          /* int(pano_h * crop_ratio) rows dropped top and bottom (:397-401); 80 */
    int32_t bev_h, bev_w;       /* BEVParams.img_h + 1, img_w + 1 (:292-293); 501, 501 */
    int32_t mask_k;             /* box-filter size of remove_hallucinated_content; 11 */
    float depth_scale;          /* uint16 depth -> metres, applied in float32 (:367); 0.001f */
    int32_t out_flags;          /* 0 for render_bev_image; 1: no vertical flip, 2: no mask (plain interpolation), 4: the renders of a
                                   launch are densified in the order given (default: the costly ones first -- same images, shorter tail).
                                   Any other bit: SALVE_ERR_BAD_ARG.  The scatter and the densify call of the same renders must be given the
                                   SAME out_flags and n: from 1025 renders per launch the scatter stage leaves a cost per render, and the
                                   densify stage its dispatch order, in the workspace's key image (which the single-render utility paths
                                   salve_bev_scatter_points / salve_bev_keys_from_pixels overwrite) */
    double win_xmin, win_xmax, win_ymin, win_ymax; /* prune_to_2d_bbox window, inclusive (:38-45); -5, 5, -5, 5 */
    double img_tx, img_ty, img_scale; /* bevimg_Sim2_world: (p + t) * s (bevparams.py:69-78); 5, 5, 50 */
    double rot_pre[4];          /* rotmat2d(-90), row-major float64 exactly as numpy computes it (:443) */
    double z_lo[2], z_hi[2];    /* crop_z_range per surface id (lo < z <= hi) (:560-566): floor, ceiling */
    double z_min;               /* z-order slicing: slices of 1 unit on [z_min, z_min + n_slices) */
    int32_t n_slices;           /* (zorder_utils.py:11: zmin -2, zmax 2, 4 slices) */
    int32_t reserved1;
} salve_bev_config_t;

/* One render = one panorama surface under one Sim(2) pose (a row of the reference's work list,
 * scripts/render_dataset_bev.py:91-109). R, t are Sim2's float32 storage (salve/common/sim2.py:50-51). */
typedef struct {
    int32_t pano_idx;   /* which panorama of the batch */
    int32_t surface;    /* 0 = floor, 1 = ceiling */
    float R[4];         /* i2Ti1.rotation, row-major */
    float t[2];         /* i2Ti1.translation (the x1.5 HoHoNet->ZInD factor is applied by the kernel, :448-451) */
    int32_t apply_pose; /* 1: pano-1 of the pair (pose applied), 0: pano-2 (identity) */
    int32_t reserved;
} salve_bev_hyp_t;

/* Bytes of device workspace salve_bev_render_batch needs for n renders (0 on bad config).  The workspace needs no
 * initialisation and carries nothing from one call to the next, except from a scatter-stage call to its densify call. */
size_t salve_bev_workspace_bytes(const salve_bev_config_t* cfg, int32_t n);

/* Panorama index: the pose-INDEPENDENT part of the rasteriser's work, built once per set of panoramas (both surfaces are
 * built) -- the counterpart of get_xyzrgb_from_depth's z filter (bev_rendering_utils.py:408-413), which the reference also
 * evaluates once per panorama and surface, before any pose is applied (:431-446).  The panorama is cut into blocks of
 * 16 x 4 pixels; the index holds, per block, the bounding box of the block's points that pass the surface's z filter, in
 * the frame after the rotmat2d(-90) product (:443-446), 16 bytes per block and surface (180 KB per panorama at 1024 x 512).
 * The render calls use it to visit, per 128 x 128 tile of the output image, only the blocks that can reach the tile under
 * the render's pose.  It must be rebuilt when the depth maps change.
 *   pano_index  device, 16-byte aligned, salve_bev_pano_index_bytes(cfg, n_panos) bytes */
size_t salve_bev_pano_index_bytes(const salve_bev_config_t* cfg, int32_t n_panos);
int salve_bev_pano_index_build(const salve_bev_config_t* cfg, const uint16_t* pano_depth, int32_t n_panos, const double* sphere,
                               void* pano_index, size_t pano_index_bytes, void* stream);

/*
 * Render n BEV texture maps.
 *   pano_rgb    device uint8  [P, pano_h, pano_w, 3]
 *   pano_depth  device uint16 [P, pano_h, pano_w]          (.depth.png payload, millimetres)
 *   sphere      device double [2*pano_h + 2*pano_w]: r[v], zdir[v], cos(theta_u), sin(theta_u), computed on the
 *               host exactly as hohonet_pano_utils.get_uni_sphere_xyz does (salve/utils/hohonet_pano_utils.py:27-43)
 *   pano_index  device: salve_bev_pano_index_build's output for exactly these P panoramas
 *   hyps        device salve_bev_hyp_t [n]
 *   out_bev     device uint32 [n, bev_h, bev_w]: final BEV image (after mask and np.flipud), 0x00BBGGRR
 *   dbg_img_xy  device int16 [n, (pano_h-2*crop_rows)*pano_w, 2] or NULL: BEV pixel (x, y) of every pano point,
 *               (-1, -1) if the point is cropped / pruned (row a4: the bit-exact index contract)
 *   dbg_keys    device uint64 [n, bev_h*bev_w] or NULL: z-order winner per pixel (unflipped):
 *               0 = empty, else ((slice+1) << 45) | (point_index << 24) | 0xBBGGRR
 *   dbg_mask    device uint8 [n, bev_h, bev_w] or NULL: hallucination mask (unflipped)
 *   dbg_stats   device int32 [n, 8] or NULL: {n_sites, min x, max x, occupied rows, walk iterations, error flag,
 *               sites handed to the general walk, queued triangles}
 *   out_in_window device int32 [n] or NULL: points inside the window per render (0 => the reference returns None, :279,
 *               and generate_texture_maps_for_pair writes no tile for the pair, :623-627)
 *   status      device int32 status word or NULL (SALVE_STATUS_*)
 */
int salve_bev_render_batch(const salve_bev_config_t* cfg, const uint8_t* pano_rgb, const uint16_t* pano_depth,
                           int32_t n_panos, const double* sphere, const void* pano_index, const salve_bev_hyp_t* hyps, int32_t n,
                           uint32_t* out_bev, int16_t* dbg_img_xy, uint64_t* dbg_keys, uint8_t* dbg_mask,
                           int32_t* dbg_stats, int32_t* out_in_window, int32_t* status, void* workspace, size_t workspace_bytes,
                           void* stream);

/* The two halves of salve_bev_render_batch as separate launches (same arguments, same workspace, same out_bev):
 * salve_bev_scatter writes the SPARSE image into out_bev (the z-order winners' colours, :307-308, already flipped) and the
 * occupancy bitmaps into the workspace; salve_bev_densify completes out_bev in place (interpolation + mask).  Used by the
 * benchmark to time the stages on their own; render_batch == scatter followed by densify. */
int salve_bev_scatter(const salve_bev_config_t* cfg, const uint8_t* pano_rgb, const uint16_t* pano_depth, int32_t n_panos,
                      const double* sphere, const void* pano_index, const salve_bev_hyp_t* hyps, int32_t n, uint32_t* out_bev,
                      int16_t* dbg_img_xy, uint64_t* dbg_keys, int32_t* out_in_window, int32_t* status, void* workspace, size_t workspace_bytes,
                      void* stream);
int salve_bev_densify(const salve_bev_config_t* cfg, int32_t n, uint32_t* out_bev, uint8_t* dbg_mask,
                      int32_t* dbg_stats, int32_t* status, void* workspace, size_t workspace_bytes, void* stream);

/* Splat an explicit coloured point cloud -- the `xyzrgb` argument of render_bev_image (bev_rendering_utils.py:254-308):
 * xyz device double [n_points, 3] in the world frame, rgb device uint8 [n_points, 3] (the reference's float colours
 * already truncated to uint8, :307-308).  The scatter stage of ONE render: writes the sparse image into out_bev
 * (uint32 [bev_h, bev_w]) and the bitmaps of render 0 into the workspace; follow with salve_bev_densify(cfg, 1, out_bev, ...).
 * n_in_window (device int32) receives the number of points inside the window (0 => render_bev_image returns None, :279). */
int salve_bev_scatter_points(const salve_bev_config_t* cfg, const double* xyz, const uint8_t* rgb, int32_t n_points,
                             uint32_t* out_bev, int32_t* n_in_window, void* workspace, size_t workspace_bytes, void* stream);

/* Stand-alone forms of the three utilities the reference exposes next to the renderer.
 * salve_zorder_winners: zorder_utils.choose_elevated_repeated_vals (salve/utils/zorder_utils.py:10-83) -- x, y device
 *   int32 [n] pixel coordinates, z device double [n], planes device double [n_slices+1] (np.linspace(zmin, zmax, ..)),
 *   scratch device uint64 [img_h*img_w], valid device uint8 [n] (1 = the point wins its pixel).
 * salve_remove_hallucinated: interpolation_utils.remove_hallucinated_content (salve/utils/interpolation_utils.py:74-122) --
 *   sparse / interp / out device uint8 [H,W,3], scratch device uint8 [H*W].
 * salve_bev_keys_from_pixels: the input side of interpolation_utils.interp_dense_grid_from_sparse (:21-54) -- xy device
 *   int32 [n,2] (x, y) pixels, rgb device uint8 [n,3]; the scatter stage of one render (sparse image into out_bev, last index
 *   wins); follow with salve_bev_densify(cfg, 1, out_bev, ...) using cfg.out_flags = 3 (no flip, no mask) to obtain the
 *   interpolated image. */
int salve_zorder_winners(const int32_t* x, const int32_t* y, const double* z, int32_t n, const double* planes, int32_t n_slices,
                         int32_t img_w, int32_t img_h, uint64_t* scratch, uint8_t* valid, void* stream);
int salve_remove_hallucinated(const uint8_t* sparse, const uint8_t* interp, int32_t H, int32_t W, int32_t K, uint8_t* scratch,
                              uint8_t* out, void* stream);
int salve_bev_keys_from_pixels(const salve_bev_config_t* cfg, const int32_t* xy, const uint8_t* rgb, int32_t n_points, uint32_t* out_bev,
                               void* workspace, size_t workspace_bytes, void* stream);

/* Panorama ingest: n RGB uint8 images [n, src_h, src_w, 3] -> [n, dst_h, dst_w, 3] with the arithmetic of
 * cv2.resize(img, (dst_w, dst_h), interpolation=cv2.INTER_LINEAR), the call every panorama goes through before
 * back-projection (salve/utils/bev_rendering_utils.py:370-375: 2048x1024 JPEG -> 1024x512).  An exact 2x down-scale
 * takes OpenCV's INTER_AREA fast path, (a + b + c + d + 2) >> 2, and needs no tables; any other size uses the 11-bit
 * fixed-point taps coef_y [dst_h, 4], coef_x [dst_w, 4] = {src0, src1, w0, w1} (same layout as salve_bev_tiles). */
int salve_resize_rgb_u8(const uint8_t* src, int32_t n, int32_t src_h, int32_t src_w, uint8_t* dst, int32_t dst_h, int32_t dst_w,
                        const int32_t* coef_y, const int32_t* coef_x, void* stream);

/* Rasterised-LAYOUT modality: n images of a filled room polygon (white) with thick anti-aliased W/D/O segments over it, flipped
 * vertically -- salve/utils/bev_rendering_utils.py:104-156 (rasterize_single_layout; cv2.fillPoly :159-179, cv2.line LINE_AA
 * :210-251).  The host has already applied the pose, the x 1.5 factor, bevimg_Sim2_world and np.round (:187-188, :214-215):
 *   layouts  device salve_layout_t [n]
 *   poly_xy  device int32 [*, 2]: polygon vertices (x, y) in pixels (closing vertex optional)
 *   segs     device int32 [*, 8]: x1, y1, x2, y2, colour 0x00BBGGRR, thickness in pixels, 0, 0
 *   out      device uint32 [n, img_h, img_w], 0x00BBGGRR (the layout of salve_bev_render_batch's out_bev: salve_bev_tiles and
 *            salve_bev_export_u8 take it as is)
 * The pixel rules are OpenCV 4.x's (modules/imgproc/src/drawing.cpp): fillPoly with its defaults for the polygon; for a segment
 * ThickLine with LINE_AA -- an anti-aliased convex quadrilateral (FillConvexPoly: LineAA along the edges, then spans) plus an
 * anti-aliased 12-gon end cap (EllipseEx / ellipse2Poly at 30 degrees) at either end, LineAA's filter and slope tables, every
 * anti-aliased pixel blended twice; thickness <= 1 is one LineAA.  LIMIT: thickness < 19 pixels (the reference draws 8- and
 * 2-pixel lines, bevparams.py:81-99); OpenCV gives thicker lines end caps at 18- / 5-degree steps, which are not implemented -- such
 * a segment is left out and SALVE_STATUS_LAYOUT_THICKNESS is raised in `status` (device int32 status word or NULL).  cv2 is not
 * installed here and the reference's tests pin none of it: the restatement is in oracle/layout_oracle.py ("parity unpinned"), the
 * kernel is bit-exact against it. */
typedef struct {
    int32_t n_poly, poly_off; /* vertex count and first vertex of this image's polygon in poly_xy */
    int32_t n_seg, seg_off;   /* segment count and first segment in segs */
} salve_layout_t;
int salve_layout_rasterise(const salve_layout_t* layouts, int32_t n, const int32_t* poly_xy, const int32_t* segs, int32_t img_h,
                           int32_t img_w, uint32_t* out, int32_t* status, void* stream);

/* BEV uint32 -> uint8 [n, bev_h, bev_w, 3], the array render_bev_image returns (bev_rendering_utils.py:328). */
int salve_bev_export_u8(const uint32_t* bev, int32_t n, int32_t bev_h, int32_t bev_w, uint8_t* out, void* stream);

/*
 * Verifier input tiles: Resize (cv2 INTER_LINEAR, uint8 fixed point) -> centre Crop -> ToTensor -> Normalize,
 * i.e. salve/train_utils.py:126-159 with salve/utils/transform.py:256-272, 386-420, 105-123, 177-202.
 *   jobs        device salve_tile_job_t [n_jobs]
 *   coef_y/x    device int32 [resize, 4]: {src0, src1, w0, w1} 11-bit taps per resized row / column
 *   lut         device float [3, 256]: (v - mean_c) / std_c evaluated in float32 on the host
 *   out         SALVE_TILE_F32_NCHW : float [slots, out_c, crop, crop]
 *               SALVE_TILE_F16_NHWC:  fp16  [slots, crop, crop, out_c]
 *               SALVE_TILE_U8X4:      uint32 [slots, crop, crop]
 */
typedef struct {
    int64_t bev_offset; /* element offset of the source image inside `bev` (uint32 units) */
    int32_t slot;       /* destination sample */
    int32_t chan;       /* first of the 3 destination channels */
} salve_tile_job_t;

#define SALVE_TILE_F32_NCHW 0
#define SALVE_TILE_F16_NHWC 1
#define SALVE_TILE_U8X4 2 /* uint32 [slots, crop, crop], 0x00BBGGRR: the Resize + Crop result itself, before ToTensor / Normalize (one
                             image per slot; chan, out_c and lut's values are not used).  For images that many hypotheses share -- the
                             identity render of a pair's second panorama -- so that salve_bev_tile_pairs need not resize them again. */

int salve_bev_tiles(const uint32_t* bev, int32_t bev_h, int32_t bev_w, const salve_tile_job_t* jobs, int32_t n_jobs,
                    const int32_t* coef_y, const int32_t* coef_x, int32_t resize, int32_t crop, const float* lut,
                    void* out, int32_t out_format, int32_t out_c, void* stream);

/* The two tiles of an early-fusion pair in ONE pass (the fused render -> verify driver's form of salve_bev_tiles, fp16 NHWC
 * only): pair k takes its first image from bev_a + jobs_a[k].bev_offset and its second from bev_b + jobs_b[k].bev_offset,
 * both go to sample jobs_a[k].slot ( == jobs_b[k].slot), channels jobs_a[k].chan .. + 2 and jobs_b[k].chan .. + 2, which
 * must be the two halves of one group of six channels (min(chan) a multiple of 6, |chan_a - chan_b| == 3: the x1 | x2 of a
 * surface, salve/models/early_fusion.py:52-60 -- either order: salve/dataset/zind_data.py:110).  A thread computes both
 * pixels and writes the six channels (and, for the last group of a sample whose out_c leaves padding channels, the zero
 * padding too) with whole-pixel stores, where two salve_bev_tiles calls write three 2-byte channels each.  Same arithmetic
 * as salve_bev_tiles: bit-identical tiles.  b_pretiled = 1: bev_b holds SALVE_TILE_U8X4 images (uint32 [*, crop, crop]) and
 * jobs_b[k].bev_offset is an element offset into THAT array: the second image of a pair is the identity render of a panorama
 * (bev_rendering_utils.py:455), the same for every hypothesis that names it -- resized and cropped once, only ToTensor +
 * Normalize per pair (the same integer taps, the same table: bit-identical again).  (int return: SALVE_ERR_BAD_ARG on null
 * pointers / bad sizes; the pairing rule is the caller's to keep.) */
int salve_bev_tile_pairs(const uint32_t* bev_a, const uint32_t* bev_b, int32_t bev_h, int32_t bev_w, const salve_tile_job_t* jobs_a,
                         const salve_tile_job_t* jobs_b, int32_t n_pairs, const int32_t* coef_y, const int32_t* coef_x, int32_t resize,
                         int32_t crop, const float* lut, void* out, int32_t out_c, int32_t b_pretiled, void* stream);

/* salve_bev_densify followed by salve_bev_tile_pairs(..., b_pretiled = 1) as ONE launch (the fused render -> verify driver's form since
 * ABI 6): every workgroup of the densify kernel, having finished its render, resizes / crops / normalises it into the verifier's sample
 * while the image is still in the L2 -- a launch of its own reads the 1 MB image back from HBM.  The job tables are indexed by RENDER
 * of the launch (render r = image r of out_bev):
 *   jobs_a[r]   .slot / .chan: destination sample and first channel of render r's tile (.bev_offset is not used); slot < 0: no tile
 *   jobs_b[r]   the pair's second image: .bev_offset = element offset of a SALVE_TILE_U8X4 image inside tiles_b, .chan its first channel
 *               (the two chans are the halves of one group of six, as for salve_bev_tile_pairs)
 * out_bev holds the complete images afterwards, exactly as after salve_bev_densify; `out` the same bits salve_bev_tile_pairs writes.
 * (bev_rendering_utils.py:254-328 + train_utils.py:126-159 / transform.py:256-272, 386-420, 105-123, 177-202.) */
int salve_bev_densify_tiles(const salve_bev_config_t* cfg, int32_t n, uint32_t* out_bev, const salve_tile_job_t* jobs_a, const salve_tile_job_t* jobs_b,
                            const uint32_t* tiles_b, const int32_t* coef_y, const int32_t* coef_x, int32_t resize, int32_t crop, const float* lut,
                            void* out, int32_t out_c, int32_t* status, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Verifier: early-fusion ResNet forward pass (fp16 MFMA, fp32 accumulation).
 * Stands behind salve/models/early_fusion.py:14-83 (EarlyFusionCEResnet), the torchvision trunk
 * selected by salve/models/resnet_factory.py:26-44, and the model build / checkpoint load of
 * salve/train_utils.py:205-242.  The host folds BatchNorm into the convolutions and describes the
 * network as a list of ops over a few activation buffers; the library executes the list.
 * ------------------------------------------------------------------------------------------------ */
#define SALVE_OP_CONV 0       /* out = relu?(conv(in) + bias (+ res)) ; NHWC fp16 */
#define SALVE_OP_MAXPOOL 1    /* 3x3 / stride 2 / pad 1 */
#define SALVE_OP_AVGPOOL_FC 2 /* global average pool + linear layer -> fp32 logits */
#define SALVE_NET_INPUT (-1)  /* buffer id of the network input */
#define SALVE_NO_BUF (-2)     /* "no residual" */

typedef struct {
    int32_t op;
    int32_t in_buf, out_buf, res_buf; /* activation buffer ids (SALVE_NET_INPUT / SALVE_NO_BUF) */
    int32_t Hi, Wi, Cin;              /* input  H, W, channels (channels padded to a multiple of 8) */
    int32_t Ho, Wo, Cout;             /* output H, W, channels (FC: Cout = number of classes) */
    int32_t KH, KW, stride, pad;      /* KW is the PADDED kernel width of the packed weights */
    int32_t relu, reserved;
    int64_t w_off;    /* CONV: element offset into the fp16 weight blob; FC: float offset of the weight in params */
    int64_t b_off;    /* float offset of the bias in params */
    int64_t ktab_off; /* CONV: offset into ktab; one int32 per 8 consecutive k: dy | dx << 8 | channel_offset << 16 */
    /* Optional second, point-wise source of a 1x1 / stride-1 CONV (in2_buf = SALVE_NO_BUF: none): the weight rows are
     * [Cin | Cin2] long and the k-tiles beyond Cin read Cin2 channels of buffer in2 (an [Hi2, Wi2, Cin2] image) at pixel
     * (oy * stride2, ox * stride2).  This is how the projection shortcut of a down-sampling bottleneck block
     * (torchvision `downsample` = 1x1 conv + BN) is folded into the block's last convolution: one GEMM over the
     * concatenated channels instead of a convolution, a tensor written and read back, and a residual add. */
    int32_t in2_buf, Cin2, stride2, Hi2, Wi2, reserved2;
} salve_resnet_op_t;

/* Kernel selection (salve_resnet_create's `flags`; 0 = the product's selection).  Every combination computes the same
 * network in the same k order with fp32 accumulation and one rounding per stored value: the logits are bit-identical, and
 * that is what these bits exist for -- the parity tests run a fused / streaming / 8-phase kernel against the plain
 * implicit-GEMM kernels it replaces.  The library reads NO environment variable (until ABI 4 it did). */
#define SALVE_RESNET_CONV_IGEMM_ONLY 1   /* conv_igemm_kernel for every convolution (no 8-phase kernel) */
#define SALVE_RESNET_CONV8_WHEREVER 2    /* the 8-phase 256 x 256 kernel wherever the shape fits, whatever the launch size */
#define SALVE_RESNET_ROUND_ROBIN_TILES 4 /* natural workgroup order instead of XCD-contiguous tiles */
#define SALVE_RESNET_NO_STEM_FUSE 8      /* 7x7 convolution and max-pool as two launches */
#define SALVE_RESNET_NO_BLOCK_FUSE 16    /* the 56 x 56 bottleneck blocks as three convolutions */
#define SALVE_RESNET_NO_PROJ_FUSE 32     /* ... only the first block of layer 1 (projection shortcut) */
#define SALVE_RESNET_NO_CHAIN 64         /* no expand_chain_kernel */
#define SALVE_RESNET_CHAIN_EXPAND_ONLY 128 /* expand_chain_kernel without the next block's first convolution */
#define SALVE_RESNET_CHAIN_16_WAVES 256  /* its 16-wave / 256-pixel-tile form for the 128-channel shapes */
#define SALVE_RESNET_CHAIN_NO_SPLIT 512  /* its 8-wave form for the 256-channel shapes too */
#define SALVE_RESNET_CHAIN_STORE_ALL 1024 /* every pixel of a stage's last block output is stored (default: only the even rows and columns
                                             that its one reader, the next stage's stride-2 projection shortcut, samples) */
#define SALVE_RESNET_NO_TRANSPOSED_TILES 2048 /* the fused 56 x 56 blocks with a fourth, half-empty tile column of 8 x 16 tiles instead of
                                             the transposed 16 x 8 tiles that cover the last 8 image columns exactly */
#define SALVE_RESNET_NO_NEXT_FUSE 4096   /* the last fused block of layer 1 without the next block's first 1x1 convolution as its fourth GEMM
                                             (with it, that block stores only the even pixels of its own output unless ..._CHAIN_STORE_ALL) */
#define SALVE_RESNET_ALL_FLAGS 8191      /* salve_resnet_create refuses any other bit */

/* Creates a handle that owns device copies of the (host) weight blobs.  NULL on failure.  flags: SALVE_RESNET_* (0). */
void* salve_resnet_create(int32_t num_layers, int32_t in_channels, const salve_resnet_op_t* ops, int32_t n_ops,
                          const void* weights_f16, size_t weights_bytes, const float* params_f32, size_t params_bytes,
                          const int32_t* ktab, size_t ktab_entries, int32_t flags);
void salve_resnet_destroy(void* handle);
int salve_resnet_num_layers(void* handle);
/* Device workspace needed for a batch (activation buffers). */
size_t salve_resnet_workspace_bytes(void* handle, int32_t batch);
/* input: device fp16 [batch, H, W, in_channels] (NHWC); logits: device float [batch, n_classes];
 * status: device int32 status word or NULL (SALVE_STATUS_FP16_RANGE). */
int salve_resnet_forward(void* handle, const void* input, int32_t batch, float* logits, void* workspace,
                         size_t workspace_bytes, int32_t* status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SALVE_HIP_H */
