/*
 * voltools_hip.h -- C ABI of the MI355X (gfx950) 3-D affine-resampling engine.
 *
 * This is the drop-in boundary for the reference's Python -> device launch interface
 * (the-lay/voltools v0.6.0).  The reference has no FFI layer of its own: its host code reaches
 * the GPU through cupy (RawKernel launches, texture objects, cp.asarray / cp.zeros / .get()).
 * Every entry point below names the reference call site it stands in for (file:line into
 * /root/reference).  The library behind it is hand-written HIP; it links against libamdhip64 only
 * (no torch, no cupy), takes plain pointers and sizes, and never throws across the boundary:
 * every function returns 0 on success, otherwise a non-zero code (a hipError_t value, or one of
 * the VT_E* codes below) and leaves a message retrievable with vt_last_error().
 *
 * Conventions
 *   - Volumes are float32, C order, shape (D, H, W); axis 0 is slowest ("depth"), axis 2 fastest.
 *   - A transform is the reference's 4x4 float32 *pull* matrix in array-axis order
 *     (transforms.py:147-152, :265-274): src[d',h',w'] = M[:3,:3] . (d,h,w) + M[:3,3].  Only the
 *     first three rows are read.  The coordinate arithmetic is carried out in float64 from those
 *     float32 (or float64, *_f64 variants) entries.
 *   - Sampling semantics are the reference GPU path's: zero border, and an output voxel is
 *     "outside" when any src+0.5 < 0 or src+0.5 >= dim (transforms.py:276-278).  Outside voxels are
 *     written as 0 unless VT_KEEP_OUTSIDE is set (then they are left untouched, as the reference's
 *     kernel does; its callers pre-zero the buffer, transforms.py:208, volume.py:73).
 *   - Handles are opaque, owned by the library, and bound to one device and one HIP stream; calls
 *     on one handle are serialised on that stream.  Output buffers are owned by the caller.
 */
#ifndef VOLTOOLS_HIP_H
#define VOLTOOLS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* interpolation= strings of the reference (transforms.py:11-17), in that order */
enum vt_interp {
    VT_LINEAR = 0,               /* 'linear'              -> linearTex3D       (helper_interpolation.h:3-6)   */
    VT_BSPLINE = 1,              /* 'bspline'             -> cubicTex3D        (helper_interpolation.h:8-40)  */
    VT_BSPLINE_SIMPLE = 2,       /* 'bspline_simple'      -> cubicTex3DSimple  (helper_interpolation.h:42-68) */
    VT_FILT_BSPLINE = 3,         /* 'filt_bspline'        -> prefilter + cubicTex3D       (transforms.py:195-197) */
    VT_FILT_BSPLINE_SIMPLE = 4   /* 'filt_bspline_simple' -> prefilter + cubicTex3DSimple                        */
};

/* flags for the *_affine calls */
enum vt_flags {
    VT_OUT_DEVICE = 1,     /* `out` is a device pointer on the handle's device (else host memory)          */
    VT_KEEP_OUTSIDE = 2,   /* leave outside voxels untouched (reference kernel: `continue`, transforms.py:278) */
    VT_FORCE_DIRECT = 4,   /* diagnostic: use the untiled global-gather kernel                              */
    VT_FORCE_TILED = 8,    /* diagnostic: use the LDS-tiled kernel even for tiny volumes                    */
    VT_NO_ZSEP = 16,       /* diagnostic: disable the axis-0-separable kernels (use the general tiled kernel) */
    VT_NO_MARCH = 32,      /* diagnostic: axis-0-separable matrices use the 3-D tiled kernel, not the marching one */
    VT_NO_ZPAIR = 64,      /* diagnostic: cubic marching on the plain layout instead of the plane-pair copy          */
    VT_NO_PACKED = 128,    /* diagnostic: general matrices use bounding-box tiles, not packed footprints             */
    VT_FORCE_PACKED = 256, /* diagnostic: packed footprints whenever they fit, even where boxes are cheaper          */
    VT_FORCE_XSWAP = 512,  /* diagnostic: rotations about axis 2 take the axis-exchange path for every interpolation  */
    VT_NO_RSWAP = 1024,    /* diagnostic: in-plane maps near a quarter turn sample the plain copy, not the transposed */
    VT_ONESHOT_EDGE_SCIPY = 4096, /* vt_affine_oneshot only: build the temporary handle with VT_EDGE_SCIPY */
    VT_NO_QUAD = 2048,     /* diagnostic: no plane-quad marching kernel (the plain / plane-pair marching kernels serve instead);
                              VT_NO_ZPAIR disables both interleaved layouts */
    VT_NO_BLOCK = 8192,    /* diagnostic: general matrices use the bounding-box / packed-footprint kernels, not the lane-block
                              kernel (VT_NO_PACKED and VT_FORCE_PACKED imply it) */
    VT_NO_ZFIR = 16384,    /* diagnostic: cubic plane-quad launches with an integer axis-0 offset keep the four-tap-plane kernel
                              instead of sampling the z-convolved copy (also what a call falls back to when that copy does not fit) */
    VT_NO_REORIENT = 32768,/* diagnostic: general matrices always sample the plain resident copy, never an axis-permuted one */
    VT_NO_ROWS = 65536     /* diagnostic: maps that leave axis 2 alone take the axis-exchange path, not the row kernel (kind 10) */
};

/* flags for vt_volume_create* */
enum vt_create_flags {
    VT_SRC_DEVICE = 1,       /* `data` is a device pointer on `dev` (else host memory)                    */
    VT_SLAB_LO_INTERIOR = 2, /* slab volumes: plane 0 of `data` is NOT the global volume's first plane     */
    VT_SLAB_HI_INTERIOR = 4, /* slab volumes: the last plane of `data` is NOT the global volume's last     */
    VT_EDGE_SCIPY = 16,      /* boundary contract of the reference's CPU path (scipy.ndimage.affine_transform, mode='constant', cval=0,
                                transforms.py:147-152) instead of the GPU path's texture contract: hard cut-off outside [0, dim-1],
                                mirrored taps inside, mirror-boundary prefilter.  Whole-volume handles only.                 */
    VT_SRC_DEFERRED = 8      /* `data` may be NULL: the resident window starts zero-filled, is filled plane range by plane
                                range with vt_volume_upload_planes and becomes usable with vt_volume_finalize              */
};

enum vt_error {
    VT_OK = 0,
    VT_EINVAL = 10001,       /* bad argument (NULL pointer, non-positive dims, unknown interpolation ...)  */
    VT_ENODEV = 10002,       /* device index out of range / no HIP device                                   */
    VT_ENOMEM = 10003,       /* host allocation failed                                                      */
    VT_EUNSUPPORTED = 10004  /* shape too large for this build's index arithmetic                           */
};

typedef struct vt_volume vt_volume_t;

typedef struct vt_volume_info {
    int32_t device;
    int32_t interp;
    int32_t depth, height, width;      /* source dims as passed to create (including any slab halo planes; the mirror padding of VT_EDGE_SCIPY handles is not counted) */
    int32_t out_depth, out_height, out_width;
    int32_t last_kernel;               /* 0 none, 1 direct, 2 tiled (boxes), 3 tiled axis-0-separable, 4 marching, 5 marching on plane pairs, 6 tiled (packed footprints), 7 fused projection (vt_volume_project), 8 marching on plane quads, 9 lane-block tiles (general matrices), 10 source rows along w (maps that leave axis 2 alone) */
    int32_t last_tile[3];              /* output tile (TD, TH, TW) of the last tiled launch (marching: G, TH, TW) */
    int32_t last_lds_dims[3];          /* staged source box (Lz, Ly, Lx) (marching: ring slots, Ly, Lx)    */
    int32_t last_lds_bytes;
    int32_t last_grid;
    float   prefilter_ms;              /* one-time prefilter time measured at create (filt_*), else 0      */
    uint64_t resident_bytes;           /* the plain resident copy and every lazily built one the handle holds now */
    float   copies_ms;                 /* GPU time spent so far building lazy copies (relayouts; hip events on the handle's stream) */
    int32_t copies_built;              /* lazy copies built so far (rebuilt ones count again) */
    int32_t copies_evicted;            /* ... and released to stay inside the budget */
    uint64_t max_resident_bytes;       /* the handle's budget (vt_volume_set_max_resident), 0 = none */
} vt_volume_info_t;

/* ---- devices: replaces general.py:61-88 (cupy.cuda.runtime.getDeviceCount, Device(i).use()) ---- */
int vt_device_count(int* count);                         /* 0 devices is not an error                    */
int vt_device_name(int dev, char* buf, int buflen);
int vt_device_props(int dev, int* cu_count, int* lds_bytes_per_block, uint64_t* hbm_bytes);
int vt_device_synchronize(int dev);

/* ---- raw device memory: replaces cp.asarray / cp.zeros / ndarray.get (transforms.py:180,223; volume.py:30,73,89) ---- */
int vt_malloc(int dev, size_t bytes, void** dptr);
int vt_free(int dev, void* dptr);
int vt_memset_zero(int dev, void* dptr, size_t bytes);
int vt_memcpy_h2d(int dev, void* dptr, const void* hptr, size_t bytes);
int vt_memcpy_d2h(int dev, void* hptr, const void* dptr, size_t bytes);
int vt_memcpy_d2d(int dev, void* dst, const void* src, size_t bytes);
/* Pin a caller-owned host range for DMA at PCIe rate (the result-buffer pool of the Python layer keeps its buffers
 * registered; the reference's `.get()`, transforms.py:223, lands in freshly allocated pageable memory). */
int vt_host_register(int dev, void* ptr, size_t bytes);
int vt_host_unregister(int dev, void* ptr);
/* Release the device buffers the library keeps for recycling (resident sources and result staging of destroyed handles
 * and one-shot calls, at most 16 GiB per device; cupy's memory pool plays this role for the reference:
 * `cp.get_default_memory_pool().free_all_blocks()`). */
int vt_device_trim(int dev);

/* ---- StaticVolume: replaces volume.py:17-59 (upload, optional prefilter, texture build; done once) ----
 * `data` holds depth*height*width float32.  For filt_* interpolations the three-pass prefilter
 * (bspline.h:30-99, launched by transforms.py:290-309) runs here, once. */
int vt_volume_create(int dev, int depth, int height, int width, int interp,
                     const float* data, int create_flags, vt_volume_t** out);

/* Slab-partitioned volume (no reference counterpart: the reference is single-GPU).  `data` holds
 * local_depth planes that are planes [plane0, plane0+local_depth) of a global volume with
 * global_depth planes; output voxels are planes [out_plane0, out_plane0+out_depth) of the global
 * output.  Source planes outside the local window read as zero. */
int vt_volume_create_slab(int dev, int local_depth, int height, int width, int interp,
                          const float* data, int create_flags,
                          int64_t plane0, int64_t global_depth, int64_t out_plane0, int out_depth,
                          vt_volume_t** out);

/* Deferred construction (multi-GPU slabs: a rank's own planes and the halo planes it receives from its neighbours land in
 * the resident buffer directly, without assembling the window in a second device buffer first).  No reference counterpart.
 * `data`: nplanes * height * width float32 (host, or device with VT_SRC_DEVICE in `flags`) for resident planes
 * [first_plane, first_plane + nplanes).  vt_volume_finalize runs the one-time prefilter (filt_*) and enables the handle. */
int vt_volume_upload_planes(vt_volume_t* vol, int first_plane, int nplanes, const float* data, int flags);
int vt_volume_finalize(vt_volume_t* vol);

/* 1 when the library is the test build that also carries round 1's kernel families (plain / plane-pair marching, kernels 4 / 5, and
 * the axis-0-separable box kernel, 3): `make LEGACY=1`.  The product build returns 0; VT_NO_QUAD / VT_NO_ZPAIR / VT_NO_MARCH then
 * send an axis-0-separable matrix to the general-matrix kernels.  (No reference counterpart: diagnostic.) */
int vt_has_legacy_kernels(void);

/* Free the resident copies the handle built lazily besides its plain one -- per orientation used: the axis-exchanged plain copy, its
 * plane-quad form and the z-convolved plane-quad form, up to 4x the volume each way round (vt_volume_info.resident_bytes) -- e.g. between
 * the phases of a job that rotate about different axes, or before another handle needs the memory.  The next call that needs a copy
 * rebuilds it; results are unchanged.  freed_bytes may be NULL.  (No reference counterpart: the reference keeps one CUDA array per
 * StaticVolume, volume.py:37-45; cupy's pool serves `free_all_blocks()` for its temporaries.) */
int vt_volume_release_copies(vt_volume_t* vol, uint64_t* freed_bytes);
/* Resident-memory budget of the handle in bytes, the plain copy included (0 = none; default: VT_MAX_RESIDENT_GB or none).  Before a lazy
 * copy is built the least recently used ones are released until it fits; a copy that cannot fit is not built and the call runs on the
 * kernel family that samples the plain layout.  Results do not depend on the budget.  (No reference counterpart: the reference keeps
 * exactly one CUDA array per StaticVolume, volume.py:37-45.) */
int vt_volume_set_max_resident(vt_volume_t* vol, uint64_t bytes);
int vt_volume_destroy(vt_volume_t* vol);
int vt_volume_info(const vt_volume_t* vol, vt_volume_info_t* info);
int vt_volume_stream(const vt_volume_t* vol, void** hip_stream);   /* the hipStream_t launches go to */
int vt_volume_sync(vt_volume_t* vol);

/* Output shape other than the source shape (scipy's output_shape, transforms.py:136-150; reshape=True). */
int vt_volume_set_output_shape(vt_volume_t* vol, int out_depth, int out_height, int out_width);

/* ---- StaticVolume.affine: replaces volume.py:61-91 (matrix upload + one kernel launch) ----
 * m4x4: 16 float32, row-major (the reference's `xform`, transforms.py:204,255).  out: out_depth *
 * out_height * out_width float32.  With VT_OUT_DEVICE the call is asynchronous on the handle's
 * stream; with a host `out` it returns after the copy back (volume.py:89). */
int vt_volume_affine(vt_volume_t* vol, const float* m4x4, float* out, int flags);
int vt_volume_affine_f64(vt_volume_t* vol, const double* m4x4, float* out, int flags);

/* ---- a batch of matrices against one resident volume (the loop of README.md:25-27 / benchmark.py:52-54 in one call)
 * m4x4s: n x 16 float32; out: n consecutive output volumes.  Volumes up to 96^3 are served by ONE kernel launch
 * (launch latency, README.md:74, is paid once); larger ones are queued back to back on the handle's stream.
 * With a host `out` the call returns after the copy back; with VT_OUT_DEVICE it returns after the launch. */
int vt_volume_affine_batch(vt_volume_t* vol, int n, const float* m4x4s, float* out, int flags);

/* ---- projection: the transformed volume summed over axis 0, without materialising it ----
 * Replaces `static_volume.transform(...).sum(axis=0)` of examples/projections.py:20-26 (a cupy reduction after the
 * kernel of volume.py:78).  out_hw: out_height * out_width float32 (host, or device with VT_OUT_DEVICE).
 * Matrices of the form [1 0 0 tz; 0 a b ty; 0 c d tx] (the example's rotation=(i,0,0) 'sxyz') are computed as one
 * weighted streaming sum of the resident planes followed by a 2-D interpolation (last_kernel 7); other matrices
 * transform into an internal buffer and sum it.  Outside voxels contribute 0; VT_KEEP_OUTSIDE is ignored. */
int vt_volume_project(vt_volume_t* vol, const float* m4x4, float* out_hw, int flags);
int vt_volume_project_f64(vt_volume_t* vol, const double* m4x4, float* out_hw, int flags);

/* ---- timing: replaces the cupy event pairs of profile=True (transforms.py:167-169,214-219; volume.py:65-67,80-85)
 * Events are recorded on the handle's stream. vt_timer_stop synchronises and returns milliseconds. */
int vt_timer_start(vt_volume_t* vol);
int vt_timer_stop(vt_volume_t* vol, float* ms);

/* ---- prefilter on a caller-owned device array: replaces _bspline_prefilter (transforms.py:290-309;
 * kernels SamplesToCoefficients3DX/Y/Z, bspline.h:58-99).  In place from the caller's point of view. */
int vt_prefilter_inplace(int dev, float* d_volume, int depth, int height, int width);

/* ---- one-shot transform(): replaces the GPU branch of affine (transforms.py:164-226):
 * host in -> upload -> (prefilter) -> kernel -> host out. */
int vt_affine_oneshot(int dev, const float* h_volume, int depth, int height, int width, int interp,
                      const float* m4x4, float* h_out, int flags, float* elapsed_ms /* may be NULL */);

const char* vt_last_error(void);
const char* vt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VOLTOOLS_HIP_H */
