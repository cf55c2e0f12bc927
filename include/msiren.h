/*
 * msiren.h -- C ABI of libmsiren.so: the MI355X (gfx950) modulated-SIREN inference path.
 *
 * Drop-in boundary for ONE path of MatteoWohlrapp/mri-inr: the dense coordinate-grid forward of
 * `ModulatedSiren` (reference: src/networks/modulated_siren.py:435-457 and everything it calls).
 * The reference has no FFI of its own -- its boundary is the Python nn.Module protocol
 * (ctor kwargs / load_state_dict / __call__) -- so every entry point below names the reference
 * interface it stands in for.  The Python mirror of that protocol (mri_inr_amd/model.py) binds
 * these symbols with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C types only; all tensors are dense row-major float32 unless stated otherwise;
 *   - every function returns 0 on success, a negative MSIREN_E_* code on failure, and leaves a
 *     human-readable message retrievable with msiren_last_error() (thread-local);
 *   - "host" pointers are ordinary process memory; "dev" pointers are HIP device memory on the
 *     handle's device (hipMalloc, msiren_dev_alloc or e.g. torch.Tensor.data_ptr());
 *   - *_dev entry points only enqueue work on the handle's stream: call msiren_sync() (or
 *     msiren_timer_stop()) before reading results;
 *   - one handle = one device + one stream + one weight set; handles are independent and may be
 *     used from different threads (a single handle is not re-entrant);
 *   - threads may hand their handles the same host arrays, or windows of one array that touch or overlap: inputs are only read, and
 *     the library never page-locks, registers or otherwise changes the state of a caller's memory (a range that the CALLER has
 *     page-locked only in part is copied through a bounce buffer: msiren_host_range_kind).  Two calls that WRITE overlapping output
 *     ranges race, as any two writers do.
 */
#ifndef MSIREN_H
#define MSIREN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 6): msiren_runtime_info, msiren_host_range_kind added; the large-call split (MSIREN_SPLIT_MIN) and the per-call page-locking of
 * caller buffers (MSIREN_HOST_REGISTER) left the library.  2 (round 5): msiren_chain_* gone, msiren_profile_read_kernel /
 * msiren_last_trunk_kernel / msiren_device_pci added; sync no longer returns MSIREN_E_RANGE.  A library of another number refuses
 * msiren_create. */
#define MSIREN_ABI_VERSION 3

#if defined(__GNUC__)
#define MSIREN_API __attribute__((visibility("default")))
#else
#define MSIREN_API
#endif

enum {
    MSIREN_OK = 0,
    MSIREN_E_INVALID = -1,  /* bad argument / unsupported configuration (Python: ValueError)  */
    MSIREN_E_STATE = -2,    /* call order: weights missing or not committed (RuntimeError)    */
    MSIREN_E_SHAPE = -3,    /* tensor size does not match the configuration (load_state_dict) */
    MSIREN_E_HIP = -4,      /* HIP runtime error (message carries hipGetErrorString)          */
    MSIREN_E_NOMEM = -5,
    MSIREN_E_RANGE = -6     /* (rounds 2-3: an operand left the domain of the split-fp16 trunk.  Not returned since round 4:
                               such launches are re-run on the exact-fp32 trunk on the stream itself, "Domain guard" below) */
};

enum { MSIREN_ACT_SINE = 0, MSIREN_ACT_MORLET = 1 };

/* arithmetic of the hidden-layer contractions */
enum {
    MSIREN_PREC_F32 = 0,  /* v_mfma_f32_32x32x2_f32: exact fp32, the parity path (configs 1-4) */
    MSIREN_PREC_BF16 = 1, /* bf16 operands, fp32 accumulate: single-product trunk (weight-stationary from 3
                             layers on), dim_hidden = 512, 2 <= num_layers <= 11: the per-layer tables must
                             fit the 160 KB LDS (BASELINE config 5; own tolerance)                    */
    MSIREN_PREC_F16X3 = 2, /* split-fp16: 3 x v_mfma_f32_16x16x32_f16 per product, fp32 accumulate;
                             fp32-equivalent accuracy (22-bit operands); H = 256, 2 <= L <= 11 (the
                             per-layer tables must fit the 160 KB LDS beside the weight ring; depths
                             3..5 run the weight-stationary kernel on single-stream handles).
                             Other shapes silently use MSIREN_PREC_F32.  A launch that meets a modulation
                             outside what fp16 operands can carry is followed, on the same stream, by the
                             exact-fp32 trunk over the same batch (a conditional launch: "Domain guard"
                             below) -- the output is the reference's fp32 result either way. */
    MSIREN_PREC_F16 = 3   /* as BF16 with fp16 operands (11-bit significand).  fp16 ends at 65 504 where the
                             reference's fp32 does not: a launch that stores a non-finite output (an overflow
                             to inf is NaN one sine later) is followed, on the same stream, by the exact-fp32
                             trunk over the same batch as a conditional launch -- outside the fp16 domain the
                             output is the fp32 trunk's, bit for bit (round 5)                          */
};

/*
 * Mirrors the keyword arguments of ModulatedSiren.__init__ (src/networks/modulated_siren.py:349-368)
 * that influence the forward pass, i.e. the `model:` block of the YAML files
 * (configuration/train_modulated_siren.yaml:14-29).  `dropout`, `modulate`, `encoder_path`
 * have no effect on eval-mode maths and stay on the Python side.
 */
typedef struct msiren_config {
    int32_t abi_version;      /* MSIREN_ABI_VERSION                                             */
    int32_t dim_in;           /* must be 2 (the grid is a 2-D meshgrid, :427-433)               */
    int32_t dim_hidden;       /* H                                                              */
    int32_t dim_out;          /* must be 1 (squeeze(2)+rearrange at :451-455)                   */
    int32_t num_layers;       /* L: number of modulated sine layers before last_layer           */
    int32_t latent_dim;       /* Z                                                              */
    float w0;                 /* frequency of layers 1..L-1 and of last_layer (:196, :211-213)  */
    float w0_initial;         /* frequency of layer 0                                           */
    int32_t use_bias;
    int32_t activation;       /* MSIREN_ACT_*; last_layer is always sine (:120-123)             */
    int32_t outer_patch_size; /* O: encoder tile (32: FixedAutoencoder is hard-wired to it)     */
    int32_t inner_patch_size; /* I: tiling stride                                               */
    int32_t siren_patch_size; /* S: output tile, P = S*S coordinates per patch                  */
    int32_t residual;         /* 0 = reference semantics; 1 = build-defined skip    */
    int32_t precision;        /* MSIREN_PREC_*                                                  */
    int32_t device;           /* HIP device ordinal                                             */
    int32_t reserved[4];
} msiren_config;

typedef struct msiren_ctx* msiren_handle;

/* ---- lifecycle ------------------------------------------------------------------------------ */

/* ModulatedSiren(**kwargs) + .to(device): validates the configuration, selects the device,
 * creates the stream.  Reference: modulated_siren.py:349-433, test_mod_siren.py:96-120. */
MSIREN_API int msiren_create(const msiren_config* cfg, msiren_handle* out);
MSIREN_API int msiren_destroy(msiren_handle h);

/* Thread-local message of the last failing call on this thread ("" if none). */
MSIREN_API const char* msiren_last_error(void);

/* ---- weights: load_state_dict (test_mod_siren.py:116-118) ------------------------------------ */

/* One state_dict entry, by its reference key name, e.g.
 *   "net.layers.0.weight" (H,2) ... "net.layers.{l}.weight" (H,H), "net.layers.{l}.bias" (H),
 *   "net.last_layer.weight" (1,H), "net.last_layer.bias" (1), "grid" (P,2),
 *   "modulator.layers.{l}.0.weight" (H,Z) / (H,H+Z), "modulator.layers.{l}.0.bias" (H),
 *   "encoder.encoder.encoder.{0,2,4}.weight/.bias", "encoder.encoder.encoder.7.weight/.bias".
 * `n` is the element count and must match the shape implied by the configuration
 * (MSIREN_E_SHAPE otherwise, like load_state_dict's size-mismatch error); unknown names are
 * MSIREN_E_INVALID ("unexpected key").  Data is copied; the caller keeps ownership. */
MSIREN_API int msiren_set_tensor(msiren_handle h, const char* name, const float* host_data, size_t n);
/* state_dict()[name]: copies the tensor the handle holds (set by msiren_set_tensor, or received by
 * msiren_broadcast_weights) into host_out; n must be its element count.  MSIREN_E_STATE if absent. */
MSIREN_API int msiren_get_tensor(msiren_handle h, const char* name, float* host_out, size_t n);

/* Packs the tensors into the kernels' layouts and uploads them.  Fails with MSIREN_E_STATE and a
 * list of missing keys if the trunk ("net.*") is incomplete; modulator / encoder keys are only
 * required by msiren_forward_latent / msiren_forward_tiles. */
MSIREN_API int msiren_commit_weights(msiren_handle h);

/* The whole state_dict as ONE flat float32 image ("blob": mri_inr_amd/csrc/weights_blob.h) -- what
 * torch.save / torch.load of the state_dict is to the reference (test_mod_siren.py:116-118) and the exact
 * payload msiren_broadcast_weights sends: header (magic, version, key count, layout hash, payload size),
 * one presence flag per key of the configuration, then every tensor (absent ones as zeros).
 *   msiren_weights_blob_size   number of floats of this configuration's blob
 *   msiren_weights_export      the tensors the handle holds -> blob_host (n_floats must be the blob size)
 *   msiren_weights_import      blob -> the handle's tensors (replacing them; keys absent in the blob stay absent)
 *                              and commit, exactly what a receiving rank of msiren_broadcast_weights executes.
 * A blob of another configuration is MSIREN_E_SHAPE, a corrupt one MSIREN_E_INVALID; the handle keeps its tensors. */
MSIREN_API int msiren_weights_blob_size(msiren_handle h, size_t* n_floats);
MSIREN_API int msiren_weights_export(msiren_handle h, float* blob_host, size_t n_floats);
MSIREN_API int msiren_weights_import(msiren_handle h, const float* blob_host, size_t n_floats);

/* ---- forward ---------------------------------------------------------------------------------- */

/* SirenNet.forward over the fixed grid (modulated_siren.py:215-233 + :448-455):
 * mods (L,B,H) -- the tuple the Modulator returns, stacked -- -> out (B,S,S).  B may be 0. */
MSIREN_API int msiren_forward_mods(msiren_handle h, const float* mods_host, int64_t B, float* out_host);
MSIREN_API int msiren_forward_mods_dev(msiren_handle h, const float* mods_dev, int64_t B, float* out_dev);

/* Modulator.forward + SirenNet.forward (modulated_siren.py:325-343): latent (B,Z) -> out (B,S,S).
 * If mods_out is non-NULL the (L,B,H) modulations are returned as well. */
MSIREN_API int msiren_forward_latent(msiren_handle h, const float* z_host, int64_t B, float* out_host, float* mods_out_host);
MSIREN_API int msiren_forward_latent_dev(msiren_handle h, const float* z_dev, int64_t B, float* out_dev, float* mods_out_dev);

/* The two producers alone, as the reference exposes them as sub-modules: `model.encoder(tiles)` -> latent (B, Z)
 * (modulated_siren.py:420, 282-301; siren_encoder.py:565-577) and `model.modulator(z)` -> the L modulation vectors, stacked
 * (L, B, H) (modulated_siren.py:416, 325-343).  msiren_forward_tiles == trunk(modulate(encode(tiles))), bit for bit. */
MSIREN_API int msiren_encode_tiles(msiren_handle h, const float* tiles_host, int64_t B, float* latent_host);
MSIREN_API int msiren_encode_tiles_dev(msiren_handle h, const float* tiles_dev, int64_t B, float* latent_dev);
MSIREN_API int msiren_modulate(msiren_handle h, const float* latent_host, int64_t B, float* mods_host);
MSIREN_API int msiren_modulate_dev(msiren_handle h, const float* latent_dev, int64_t B, float* mods_dev);
/* ModulatedSiren.forward (modulated_siren.py:435-457), custom-encoder branch
 * (siren_encoder.py:503-512,565-577): tiles (B,O,O) -> out (B,S,S). */
MSIREN_API int msiren_forward_tiles(msiren_handle h, const float* tiles_host, int64_t B, float* out_host);
MSIREN_API int msiren_forward_tiles_dev(msiren_handle h, const float* tiles_dev, int64_t B, float* out_dev);

/* The slice pipeline around the model call as metrics_error drives it (src/util/error.py:231-249):
 * image (Hh,Ww) -> image_to_patches(O,I) (tiling.py:10-64) -> black-patch filter (mean < 1e-10,
 * tiling.py:184-198,244-271) -> ModulatedSiren.forward on the non-black tiles -> zeros re-inserted
 * (tiling.py:274-303) -> weighted overlap-add (tiling.py:67-140) -> recon (nV*I, nH*I).
 * n_slices images of identical size are processed as one batch.  recon_rows/cols may be NULL.
 * The host-pointer form stores the reconstruction straight into recon_host where that is page-locked memory (see msiren_host_alloc below). */
MSIREN_API int msiren_reconstruct_slices_dev(msiren_handle h, const float* images_dev, int64_t n_slices, int32_t height,
                                  int32_t width, float* recon_dev);
MSIREN_API int msiren_reconstruct_slices(msiren_handle h, const float* images_host, int64_t n_slices, int32_t height,
                              int32_t width, float* recon_host);
/* The same chain from tiles that are already cut (what metrics_error receives, error.py:200-249):
 * tiles (n*nV*nH, O, O) -> black filter -> model -> zeros re-inserted -> weighted overlap-add -> (n, nV*I, nH*I). */
MSIREN_API int msiren_reconstruct_tiles_dev(msiren_handle h, const float* tiles_dev, int64_t n_slices, int32_t n_vertical,
                                 int32_t n_horizontal, float* recon_dev);
/* Output geometry of the above: nV = ceil(height/I), nH = ceil(width/I); recon is (nV*I, nH*I). */
MSIREN_API int msiren_recon_shape(msiren_handle h, int32_t height, int32_t width, int32_t* n_vertical, int32_t* n_horizontal);

/* Stand-alone tiling steps on device buffers (same references as above). */
MSIREN_API int msiren_image_to_patches_dev(msiren_handle h, const float* images_dev, int64_t n_slices, int32_t height,
                                int32_t width, float* patches_dev /* (n*nV*nH, O, O) */);
MSIREN_API int msiren_weighted_fold_dev(msiren_handle h, const float* tiles_dev /* (n*nV*nH, S, S) */, int64_t n_slices,
                             int32_t n_vertical, int32_t n_horizontal, float* recon_dev);

/* patches_to_image (tiling.py:143-181): plain overlap average fold(tiles) / fold(ones) of O x O tiles at stride I,
 * padding (O-I)/2 -> (n, nV*I, nH*I).  metrics_error folds the fully-sampled and the undersampled tiles with it
 * to get the images it scores against (error.py:250-255). */
MSIREN_API int msiren_patches_to_image_dev(msiren_handle h, const float* tiles_dev /* (n*nV*nH, O, O) */, int64_t n_slices,
                                int32_t n_vertical, int32_t n_horizontal, float* image_dev);
/* The black-patch filter of the reference's callers as separate steps (src/util/tiling.py:184-198, 244-303; used one by one in
 * src/train/training.py:438-445 -- msiren_reconstruct_tiles_dev does all of them in one call).  All on the handle's current
 * stream; tiles / rows float32, flags / indices int32, everything device memory.
 *   msiren_black_patch_flags_dev  flags[t] = 1 where mean(tile t) < 1e-10 (classify_patches; a tile of 1e-12 is black), else 0
 *   msiren_gather_rows_dev        dst[j] = src[idx[j]], j < n_idx            (patches[non_black_indices])
 *   msiren_scatter_rows_dev       dst = zeros(n_rows); dst[idx[j]] = src[j]  (reintegrate_black_patches: black rows stay zeros) */
MSIREN_API int msiren_black_patch_flags_dev(msiren_handle h, const float* tiles_dev, int64_t n_tiles, int64_t tile_elems, int32_t* flags_dev);
MSIREN_API int msiren_gather_rows_dev(msiren_handle h, const float* src_dev, const int32_t* idx_dev, int64_t n_idx, int64_t row_elems, float* dst_dev);
MSIREN_API int msiren_scatter_rows_dev(msiren_handle h, const float* src_dev, const int32_t* idx_dev, int64_t n_idx, int64_t n_rows, int64_t row_elems,
                                       float* dst_dev);

/* Pipelining of asynchronous calls.  n = 1 (default): every *_dev call is enqueued on one stream and
 * executes in call order.  n = 2: consecutive *_dev FORWARD calls (forward_mods/latent/tiles_dev,
 * reconstruct_slices_dev) alternate between two streams with private scratch, so independent calls
 * overlap on the device (the under-occupied tail of one call's trunk kernel is filled by the next
 * call's kernels).  n = 3 (round 5): a rotation over three -- call k+2's encoder / Modulator no longer queue behind call k's
 * trunk, which pays where the trunk OWNS its CUs (config 5: +3.7 %; the default model: +0.1 %).  The caller then must not hand the
 * same output buffer to n consecutive calls, nor feed one call's output to the next, without an msiren_sync() in between.
 */
MSIREN_API int msiren_set_streams(msiren_handle h, int32_t n);

/* Blocks until everything enqueued on the handle's streams has finished (the reference's implicit
 * synchronisation at .cpu(), error.py:256-258). */
MSIREN_API int msiren_sync(msiren_handle h);

/* ---- multi-GPU: weights replicated by ONE RCCL broadcast, patches sharded by the caller ------------
 *
 * The reference is single-process, single-GPU (practical_slurm_launcher.sh:8-11, test_mod_siren.py:90-93);
 * patches are independent given the weights, so the forward has no exchange step and the only
 * collective of the scale-out is the broadcast of the state_dict from one rank at load time
 * (load_state_dict, test_mod_siren.py:116-118, executed on one rank instead of all).  librccl is
 * dlopen'ed by the first of these calls: a single-GPU host never loads it.
 *
 * One process per GPU:   rank 0: msiren_comm_unique_id(id) -> ship the 128 bytes to the other ranks
 *                        all:    msiren_comm_init_rank(h, id, 128, nranks, rank)
 *                        all:    msiren_broadcast_weights(h, root)      [collective; commits the weights]
 * One process, n GPUs:   msiren_comm_init_all(handles, n); msiren_broadcast_weights_all(handles, n, root)
 *
 * msiren_broadcast_weights: the root must hold every tensor it wants replicated (msiren_set_tensor);
 * the key set travels with the payload, the other ranks end up with exactly the root's tensors and
 * every rank's weights are committed (as by msiren_commit_weights) when the call returns.
 * msiren_comm_barrier / msiren_comm_allreduce_max_f64 are the two plumbing collectives a benchmark
 * needs (barrier around the timed region, MAX of the per-rank times); both synchronise the host. */
#define MSIREN_COMM_ID_BYTES 128
MSIREN_API int msiren_comm_unique_id(void* id_out, size_t bytes);
MSIREN_API int msiren_comm_init_rank(msiren_handle h, const void* id, size_t bytes, int32_t nranks, int32_t rank);
MSIREN_API int msiren_comm_init_all(msiren_handle* handles, int32_t n);
MSIREN_API int msiren_broadcast_weights(msiren_handle h, int32_t root);
MSIREN_API int msiren_broadcast_weights_all(msiren_handle* handles, int32_t n, int32_t root);
MSIREN_API int msiren_comm_barrier(msiren_handle h);
MSIREN_API int msiren_comm_allreduce_max_f64(msiren_handle h, double* inout, int32_t n);
MSIREN_API int msiren_comm_info(msiren_handle h, int32_t* nranks, int32_t* rank);
MSIREN_API int msiren_comm_destroy(msiren_handle h);

/* ---- device memory + timing helpers (so that a host needs no other GPU runtime) --------------- */

MSIREN_API int msiren_dev_alloc(msiren_handle h, size_t bytes, void** dev_ptr);
MSIREN_API int msiren_dev_free(msiren_handle h, void* dev_ptr);
/* Host buffers of the host-pointer entry points.  Where a caller's buffer is page-locked memory -- from here, or any memory the HIP
 * runtime has page-locked: a torch tensor after .pin_memory(), what the reference's DataLoader delivers with pin_memory=True -- the kernels of
 * a msiren_forward_tiles call of fewer than 2400 tiles work on it IN PLACE (the trunk stores into the output array, the conv kernel reads the
 * tiles), and msiren_reconstruct_slices stores the reconstruction into it; ordinary pageable memory is copied by the runtime.  The Python
 * mirror takes its OUTPUT arrays from a bounded recycling pool of these blocks by default: one 320x320 slice numpy -> numpy 485 (round 4) ->
 * 380 us, 364 with page-locked tiles as well.  Larger calls cut themselves into chunks over the handle's two streams, with copies that run
 * beside the other chunk's kernels.  Same results either way, bit for bit.
 * The library itself never calls hipHostRegister / hipHostUnregister on a caller's memory (round 5 did so per call for a few hours; the
 * path was deleted in round 6 after an unexplained GPU memory fault in processes that used it: profiles/r6/01_*).
 * msiren_host_range_kind: what a host range is to the entry points above -- 0 = pageable (copied by the runtime), 1 = the whole range lies
 * inside ONE page-locked allocation (used in place), 2 = page-locked in part (it begins or ends inside a page-locked allocation that does
 * not hold all of it: goes through a bounce buffer).  No handle: any thread, any time after the first HIP call of the process. */
MSIREN_API int msiren_host_alloc(msiren_handle h, size_t bytes, void** host_ptr);
MSIREN_API int msiren_host_free(msiren_handle h, void* host_ptr);   /* h may be NULL: a block that has outlived its handle */
MSIREN_API int msiren_host_range_kind(const void* host_ptr, size_t bytes, int32_t* kind);
MSIREN_API int msiren_memcpy_h2d(msiren_handle h, void* dst_dev, const void* src_host, size_t bytes);
MSIREN_API int msiren_memcpy_d2h(msiren_handle h, void* dst_host, const void* src_dev, size_t bytes);

/* HIP events on the handle's stream: start .. stop brackets the launches enqueued in between;
 * stop synchronises and returns elapsed milliseconds. */
MSIREN_API int msiren_timer_start(msiren_handle h);
MSIREN_API int msiren_timer_stop(msiren_handle h, float* elapsed_ms);
/* Per-kernel accounting: while enabled, every launch of the fused trunk kernel is bracketed by its
 * own event pair; msiren_profile_read returns launch count and summed milliseconds since enable. */
MSIREN_API int msiren_profile_enable(msiren_handle h, int32_t on);
MSIREN_API int msiren_profile_read(msiren_handle h, int64_t* launches, double* trunk_ms_total);
/* The same, per trunk instance: entry `index` (0-based, in order of first launch since msiren_profile_enable(h, 1)) ->
 * its name as launched (e.g. "siren_trunk_f16x3w_kernel<0,4>"), launch count, summed milliseconds and the coordinates
 * (patches x siren_patch_size^2) its launches evaluated -- a host call of several slices runs two trunk instances, so a roofline figure is per instance: msiren_flops_per_coord x coords_total / ms_total.  MSIREN_E_INVALID past
 * the last entry.  msiren_last_trunk_kernel: the instance the most recent trunk launch of the handle used. */
MSIREN_API int msiren_profile_read_kernel(msiren_handle h, int32_t index, char* name128, int64_t* launches, double* ms_total,
                                          int64_t* coords_total);
MSIREN_API int msiren_last_trunk_kernel(msiren_handle h, char* name128);

/* name (<=255 chars + NUL), compute units, clock in MHz, total HBM bytes of the handle's device. */
MSIREN_API int msiren_device_info(msiren_handle h, char* name256, int32_t* compute_units, int32_t* clock_mhz,
                       uint64_t* hbm_bytes);
/* PCI bus id of the handle's device, "0000:c1:00.0" (<= 31 chars + NUL): which physical card a rank of a multi-GPU job sits on. */
MSIREN_API int msiren_device_pci(msiren_handle h, char* busid32);
MSIREN_API int msiren_device_count(int32_t* count);
/* Which HIP runtime the library's calls are bound to IN THIS PROCESS: *runtime_version = hipRuntimeGetVersion() (e.g. 70253625),
 * *built_against = the HIP_VERSION libmsiren.so was compiled with, *driver_version = hipDriverGetVersion(), lib_path = the file the
 * dynamic loader mapped for libamdhip64 (dladdr of a HIP entry point).  The library links libamdhip64.so.7 by soname; a PyTorch-ROCm
 * wheel bundles a libamdhip64.so of the same soname, so in a process that imported torch FIRST (the reference's own host program:
 * test_mod_siren.py:1-20 imports torch before anything else) every HIP call of this library runs on torch's bundled runtime, in a
 * torch-free process on the system one.  Any pointer may be NULL.  Needs no handle and no device. */
MSIREN_API int msiren_runtime_info(int32_t* runtime_version, int32_t* built_against, int32_t* driver_version, char* lib_path,
                                   size_t lib_path_bytes);
/* Domain guard of the split-fp16 trunk (MSIREN_PREC_F16X3).  Its fp16 operands carry activation x modulation x the next
 * layer's power-of-two weight scale; the weights are scaled into range at commit, a modulation cannot be known before the
 * call.  Every f16x3 trunk launch checks the scaled modulations it stages; if one exceeds 65504 (or is not finite) it
 * writes its launch number to a word in device memory.  Behind every such launch -- host-pointer and *_dev entry points
 * alike, one stream or two -- the library enqueues the exact-fp32 trunk over the same batch and output buffer as a
 * CONDITIONAL launch on the same stream: its workgroups read the word first and leave (~2 us) unless it holds that number.
 * So the output buffer always ends up holding what the reference's fp32 arithmetic computes (modulated_siren.py:215-233)
 * -- identical semantics, no error to handle, nothing invalidated.  msiren_range_events: synchronising calls that found a
 * conditional launch had run, since msiren_create (informational: such a model is better served by MSIREN_PREC_F32). */
MSIREN_API int msiren_range_events(msiren_handle h, int64_t* count);
/* Diagnostic: the rate the device sustains on nothing but the split-fp16 trunk's MFMA stream (v_mfma_f32_16x16x32_f16, one wave
 * per SIMD on every CU, the trunk's three products per k-step on operands of the trunk's magnitudes), ~10 ms.  *tflops: fp16
 * MFMA TFLOP/s issued chip-wide (divide by 3 for the algorithmic figure of the f16x3 roofline); *mhz_equivalent (optional):
 * the clock at which one MFMA per 16 cycles and SIMD gives that rate.  What the nominal peak becomes under the power limit
 * on real data; bench.py reports it beside the roofline, never as `peak`. */
MSIREN_API int msiren_mfma_sustained_probe(msiren_handle h, double* tflops, double* mhz_equivalent);
/* Algorithmic FLOPs per coordinate for the handle's configuration: 2*2*H + (L-1)*2*H*H + 2*H. */
MSIREN_API int msiren_flops_per_coord(msiren_handle h, double* flops);
MSIREN_API int msiren_abi_version(void);
/* Diagnostic (H=256, sine only): runs a stamped build of the trunk kernel once and returns, per
 * workgroup, 32 uint64: [0] HW_ID, [1] LDS_ALLOC, [2] XCC_ID, [3] s_memrealtime at start,
 * [4..] s_memtime at each phase boundary.  Never used by the forward entry points. */
/* Same for the register-resident f16x3 trunk: per workgroup and pass (first 4), 48 uint64: [0] s_memtime at pass
 * start, [1] after layer 0, [2] after the hidden layers, [6] at pass end, [7] s_memrealtime at pass end, [8..39]
 * s_memtime at the end of each of the 32 hidden-layer tiles.  (H=256, L=5, sine.) */
MSIREN_API int msiren_f16x3_timeline(msiren_handle h, const float* mods_dev, int64_t B, float* out_dev,
                                     uint64_t* stamps_host);
/* Same for the weight-stationary f16x3 trunk: per workgroup and slot (first 96), 8 uint64: [0..2] s_memtime at the top of the
 * slot's bookkeeping, at the start and at the end of its MFMA body, [3] s_memrealtime at the end, [4..6] s_memtime inside the slot
 * boundary (modulation rows staged / pass id handled / end).  (H=256, sine.) */
MSIREN_API int msiren_f16x3w_timeline(msiren_handle h, const float* mods_dev, int64_t B, float* out_dev,
                                      uint64_t* stamps_host);
MSIREN_API int msiren_trunk_timeline(msiren_handle h, const float* mods_dev, int64_t B, float* out_dev,
                                     uint64_t* stamps_host);

#ifdef __cplusplus
}
#endif
#endif /* MSIREN_H */
