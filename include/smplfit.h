/*
 * smplfit.h — C-ABI of the MI355X-native SMPL-family fitter (libsmplfit_hip.so).
 *
 * Drop-in boundary for ONE hot path of isarandi/smplfitter: smplfitter.pt.BodyFitter.fit()
 * (reference src/smplfitter/pt/bodyfitter.py:283-549) and the LBS forward it is scored with
 * (reference src/smplfitter/pt/bodymodel.py:121-307).  The reference is pure Python/PyTorch and has
 * no FFI of its own; these entry points are what a ctypes binding inside its BodyFitter /
 * BodyModel would call (see INTEGRATION.md for that stub).  Plain pointers and sizes only — no
 * torch types.  All array arguments of the compute calls are DEVICE pointers (HIP, gfx950), fp32,
 * C-contiguous; all work is enqueued on the caller's stream; nothing synchronises the device.
 *
 * Ownership: the handle owns the uploaded model constants and derived tables only.  The caller
 * owns inputs, outputs and the workspace (size from smplfit_workspace_bytes).  A handle is
 * read-only after creation: concurrent calls on different streams are safe if their workspaces
 * are distinct.
 *
 * Errors: every call returns 0 on success or a negative smplfit_status; smplfit_last_error()
 * returns a thread-local message.  No exceptions cross the boundary.  A non-SPD Gramian yields NaN
 * outputs (the reference discards cholesky_ex's info, pt/bodyfitter.py:1083).
 */
#ifndef SMPLFIT_H_
#define SMPLFIT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smplfit_handle smplfit_handle;

typedef enum smplfit_status {
  SMPLFIT_OK = 0,
  SMPLFIT_ERR_BAD_ARG = -1,     /* null pointer, bad size, option conflict                 */
  SMPLFIT_ERR_UNSUPPORTED = -2, /* model shape / option outside what the kernels implement */
  SMPLFIT_ERR_WORKSPACE = -3,   /* workspace too small / misaligned                         */
  SMPLFIT_ERR_HIP = -4,         /* HIP runtime error (no device, launch failure, OOM)       */
} smplfit_status;

/* Model constants as the reference's BodyModel holds them (pt/bodymodel.py:80-93), HOST pointers,
 * fp32, C-contiguous, original vertex order.  Copied during smplfit_create. */
typedef struct smplfit_model_desc {
  int32_t num_vertices;            /* V                                                        */
  int32_t num_joints;              /* J (<= 64)                                                */
  int32_t num_betas;               /* S                                                        */
  int32_t is_smpl_family;          /* model_name.startswith('smpl') (pt/bodyfitter.py:34)      */
  const float* v_template;         /* (V,3)  identity-pose corrective already folded in        */
  const float* shapedirs;          /* (V,3,S)                                                  */
  const float* posedirs;           /* (V,3,9(J-1))                                             */
  const float* weights;            /* (V,J) dense skinning weights                             */
  const float* J_template;         /* (J,3)                                                    */
  const float* J_shapedirs;        /* (J,3,S)                                                  */
  const int32_t* parents;          /* (J) kinematic parents, parents[0] ignored                */
  const float* J_regressor_post_lbs; /* (J, regressor_num_vertices) or NULL                    */
  int32_t regressor_num_vertices;  /* must equal V for the joints-omitted path                 */
  /* kid blend shape (BodyFitter(enable_kid=True), pt/bodyfitter.py:52-58; BodyModel.forward's
   * kid_factor, pt/bodymodel.py:258-295): when enable_kid != 0 the handle carries ONE extra shape
   * unknown whose vertex / joint directions are kid_shapedir / kid_J_shapedir. */
  int32_t enable_kid;
  const float* kid_shapedir;       /* (V,3) or NULL                                            */
  const float* kid_J_shapedir;     /* (J,3) or NULL                                            */
} smplfit_model_desc;

enum { SMPLFIT_CREATE_HOST_ONLY = 1 }; /* build tables, upload nothing (no GPU needed)         */

/* Builds the part / level / sparse-skinning tables of BodyFitter.__init__
 * (pt/bodyfitter.py:25-233) and uploads model + tables to the current HIP device. */
int smplfit_create(const smplfit_model_desc* desc, int flags, smplfit_handle** out);
void smplfit_destroy(smplfit_handle* h);

const char* smplfit_last_error(void);
const char* smplfit_version(void);
/* Version of this header's structs and entry points; bumped whenever a struct gains a field or a signature
 * changes.  A client compares it with the SMPLFIT_ABI_VERSION it was built against before the first call. */
#define SMPLFIT_ABI_VERSION 5
int smplfit_abi_version(void);

typedef struct smplfit_info {
  int32_t num_vertices, num_joints, num_betas; /* num_betas excludes the kid unknown */
  int32_t has_kid;
  int32_t padded_vertices;     /* Vp: vertices padded for the kernels                          */
  int32_t num_used_vertices;   /* vertices entering the part sums (pt/bodyfitter.py:109-114)   */
  int32_t skin_width;          /* (joint, weight) pairs kept per vertex: 4, 8, or — general path — the largest
                                  count of non-zero weights of a vertex rounded up to 4 (up to 64)              */
  int32_t num_segments;        /* part-aligned 64-vertex tiles                                 */
  int32_t num_fk_levels;
  int32_t adj_last_level;
  int32_t has_device;
  int32_t gemm_vgprs;          /* registers per lane of the split-bf16 GEMM kernels as built (0 without a device):
                                  they must own whole CUs (>= 256); below that the fp32-MFMA GEMM runs instead  */
  int32_t vertex_path;         /* which kernels the vertex passes of a default (unit-weight) fit run on:
                                  SMPLFIT_PATH_BATCH_MAJOR (lane = instance, the fast path: <= 8 weights per vertex,
                                  10 or 16 betas with or without the kid unknown, >= 1024 vertices),
                                  SMPLFIT_PATH_WAVE (one wave per instance: small subsets, non-normalised weights —
                                  about 0.4x the rate) or SMPLFIT_PATH_GENERAL (any number of betas up to 1023 / of
                                  skinning weights up to 64: run-time loops, the normal equations as one rank-k
                                  update on the matrix cores, every option of the fit)              */
  int32_t share_fallback;      /* bit k: cell table k (see smplfit_get_share_table) is a copy of a wider or coarser
                                  one because its own domain was too small for its cells; 0xffff: the model has
                                  no cell tables at all (SMPLFIT_PATH_WAVE)                                        */
} smplfit_info;
enum smplfit_vertex_path { SMPLFIT_PATH_WAVE = 0, SMPLFIT_PATH_BATCH_MAJOR = 1, SMPLFIT_PATH_GENERAL = 2 };
int smplfit_get_info(const smplfit_handle* h, smplfit_info* info);

/* Introspection of the host tables, for tests.  Copies up to cap int32 entries, sets *n. */
enum smplfit_table_id {
  SMPLFIT_TAB_PART_ASSIGNMENT = 0, /* (V)  argmax weight, toes->feet                            */
  SMPLFIT_TAB_SORT_PERM = 1,       /* (Vp) original index of sorted slot, -1 = padding          */
  SMPLFIT_TAB_PART_TYPE = 2,       /* (J)  0 none, 1 multi-joint, 2 bone, 3 leaf                */
  SMPLFIT_TAB_FK_ORDER = 3,        /* joints by tree level (pt/bodyfitter.py:181-192)           */
  SMPLFIT_TAB_FK_LEVEL_START = 4,  /* (levels+1)                                                */
  SMPLFIT_TAB_ADJ_FLAG = 5,        /* (J)  1 if refined by the final adjustment                 */
  SMPLFIT_TAB_USED_PART = 6,       /* (J)  1 if the part's vertices enter the part sums         */
  SMPLFIT_TAB_SEGMENTS = 7,        /* (nseg,3) start, count, part                               */
  SMPLFIT_TAB_VERTEX_PIECES = 8,   /* (npieces,5) start, count, part, used, joints: the pieces of the
                                      sorted slots the batch-major vertex kernels walk (runs of one
                                      part with at most four skinning joints)                    */
  SMPLFIT_TAB_JOINT_PAIRS = 10,    /* (npairs,2) joints j1 < j2 that share a vertex: the units of the pair-Gram form  */
  SMPLFIT_TAB_CELL_COUNTS = 9,     /* (8) cells per instance block of the cell tables below: the four kinds'
                                      coarse tables, then their fine ones (empty: the model has no
                                      batch-major tables)                                          */
};
int smplfit_get_table(const smplfit_handle* h, int table_id, int32_t* dst, size_t cap, size_t* n);

/* Cell tables of the batch-major vertex kernels (test access): `kind` 0 residual pass, 1 part sums over every slot,
 * 2 over the used parts, 3 over the adjustable parts; 4-7: the FINE tables of the same kinds, which batches of up
 * to SMPLFIT_FINE_B (default 768) instances walk — eight times the cells, a short walk per wave; the rows of partial
 * sums differ, so a small batch agrees with a large one to rounding, not bit for bit.
 * `what` 0 piece_start (ncells + 1), 1 piece records
 * (npieces + 1, 12): count, joints[4], local slots[4], first slot, row closed behind the piece (-1 none), kind 0:
 * joints of that row | (cell + 1) << 8 behind the last piece of a cell; 2 the rows: part per row (kinds 1-3) or
 * (nrows, 12) joint of every local slot, -1 unused (kind 0). */
int smplfit_get_share_table(const smplfit_handle* h, int kind, int what, int32_t* dst, size_t cap, size_t* n);
/* Cells per wave a launch over `batch` instances picks for `kind` 0-3 — of the fine table when the batch takes it
 * (with the current tuning options). */
int smplfit_pick_share_mult(const smplfit_handle* h, int kind, int batch);

/* Bytes of device workspace a call on `batch` instances needs (256-byte aligned base). */
size_t smplfit_workspace_bytes(const smplfit_handle* h, int batch);

/*
 * BodyFitter.fit, default configuration (pt/bodyfitter.py:283-549): no share_beta, no scale, no kid,
 * no warm start.
 *   target_vertices (B,V,3); target_joints (B,J,3) or NULL; vertex_weights (B,V) or NULL;
 *   joint_weights (B,J) or NULL.
 * Outputs: pose_rotvecs (B,3J), shape_betas (B,S), trans (B,3); kid_factor (B) (only for kid
 * handles, else NULL), orientations (B,J,3,3) and relative_orientations (B,J,3,3) (parent^T @
 * global, pt/bodyfitter.py:523-533) may be NULL.  kid_regularizer: ridge weight of the kid unknown
 * (the reference defaults it to beta_regularizer, pt/bodyfitter.py:1235-1237); ignored without kid.
 */
int smplfit_fit_f32(const smplfit_handle* h, const float* target_vertices,
                    const float* target_joints, const float* vertex_weights,
                    const float* joint_weights, int batch, int num_iter, float beta_regularizer,
                    float beta_regularizer2, float kid_regularizer, int final_adjust_rots,
                    float* pose_rotvecs, float* shape_betas, float* trans, float* kid_factor,
                    float* orientations, float* relative_orientations, void* workspace,
                    size_t workspace_bytes, void* hip_stream);

/*
 * BodyModel.forward (pt/bodymodel.py:121-307).  Exactly one of pose_rotvecs (B,3J) /
 * glob_rotmats (B,J,3,3) non-NULL; shape_betas (B,num_betas_given) or NULL; trans (B,3) or NULL;
 * kid_factor (B) or NULL (kid handles only).
 * Outputs: vertices (B,V,3) may be NULL (joints only); joints (B,J,3); orientations (B,J,3,3) may
 * be NULL.
 */
int smplfit_forward_f32(const smplfit_handle* h, const float* pose_rotvecs,
                        const float* glob_rotmats, const float* shape_betas, int num_betas_given,
                        const float* trans, const float* kid_factor, int batch, float* vertices,
                        float* joints,
                        float* orientations, void* workspace, size_t workspace_bytes,
                        void* hip_stream);

/* The general form of smplfit_forward_f32: additionally rel_rotmats (B,J,3,3), the relative rotation matrices of
 * BodyModel.forward (pt/bodymodel.py:230-234: global rotations by the kinematic chain, inside the joint kernel).
 * Exactly one of pose_rotvecs / glob_rotmats / rel_rotmats non-NULL (all NULL: rest pose).  Zero-initialise. */
typedef struct smplfit_forward_args {
  const float* pose_rotvecs;   /* (B,3J) or NULL */
  const float* glob_rotmats;   /* (B,J,3,3) or NULL */
  const float* rel_rotmats;    /* (B,J,3,3) or NULL */
  const float* shape_betas;    /* (B,num_betas_given) or NULL */
  int32_t num_betas_given;
  const float* trans;          /* (B,3) or NULL */
  const float* kid_factor;     /* (B) or NULL */
  int32_t batch;
  float* vertices;             /* out (B,V,3) or NULL */
  float* joints;               /* out (B,J,3) */
  float* orientations;         /* out (B,J,3,3) or NULL */
  void* workspace;
  size_t workspace_bytes;
  void* hip_stream;
} smplfit_forward_args;
int smplfit_forward_ex_f32(const smplfit_handle* h, const smplfit_forward_args* args);

/* BodyFitter.fit with a warm start (pt/bodyfitter.py:363-382): smplfit_fit_f32 plus
 *   initial_pose_rotvecs (B,3J) or NULL, initial_shape_betas (B,num_initial_betas) or NULL,
 *   initial_kid_factor (B) or NULL (enable_kid handles only).
 * When a pose or betas are given, the first rotation pass runs against the model posed with the initial
 * values (composed with its orientations) instead of the template, and the ridge of every shape solve
 * pulls towards initial_shape_betas / initial_kid_factor (beta/kid_regularizer_reference, :1072-1081,
 * :1224-1255).  All three NULL is exactly smplfit_fit_f32.  This is the call BodyFlipper.flip makes
 * (pt/bodyflipper.py:71-81). */
int smplfit_fit_warm_f32(const smplfit_handle* h, const float* target_vertices,
                         const float* target_joints, const float* vertex_weights,
                         const float* joint_weights, int batch, int num_iter, float beta_regularizer,
                         float beta_regularizer2, float kid_regularizer, int final_adjust_rots,
                         const float* initial_pose_rotvecs, const float* initial_shape_betas,
                         int num_initial_betas, const float* initial_kid_factor, float* pose_rotvecs,
                         float* shape_betas, float* trans, float* kid_factor, float* orientations,
                         float* relative_orientations, void* workspace, size_t workspace_bytes,
                         void* hip_stream);

/* The general form of the fit call: every option of the two entry points above in one struct, plus
 *   share_beta != 0 -- BodyFitter.fit(share_beta=True): one shape (betas [+ kid]) for the whole batch;
 *   the regularised, centred normal equations of all instances are summed before the Cholesky solve
 *   (pt/lstsq.py:24-26 through lstsq_partial_share :32-90 with every unknown shared), every instance
 *   keeps its own translation.  The sum runs over the instances in order on one GPU; when the batch
 *   is sharded over ranks (one process per GPU) `share_allreduce` completes it: after the local sum
 *   of every shape solve the driver calls it once, on the calling thread, and it must ENQUEUE on
 *   `hip_stream` an in-place sum all-reduce of `sums` (device memory, `count` = S*S + S doubles) over
 *   the ranks -- e.g. ncclAllReduce(sums, sums, count, ncclDouble, ncclSum, comm, stream) -- and
 *   return 0.  Every rank then solves the same summed system.  NULL = the batch is the whole batch.
 *   scale_mode -- BodyFitter.fit(scale_target=True) / (scale_fit=True): the LAST shape solve gets one more
 *   unknown, the refinement sees scaled targets resp. a scaled reference, the mean is added back scaled.
 *   shape_betas / kid_factor are returned as the reference returns them (undivided by the scale).
 *   share_beta together with scale_mode: the scaled solve shares the shape and keeps one scale per
 *   instance (lstsq_partial_share with n_shared = S, pt/lstsq.py:50-90): every instance contributes the
 *   Schur complement of its scale entry, the S x S sums are solved once, the scales follow.
 * Zero-initialise the struct; fields left 0 / NULL mean "not given". */
typedef int (*smplfit_share_allreduce_fn)(void* user, double* sums, int32_t count, void* hip_stream);

typedef struct smplfit_fit_args {
  const float* target_vertices;      /* (B,V,3) */
  const float* target_joints;        /* (B,J,3) or NULL */
  const float* vertex_weights;       /* (B,V) or NULL */
  const float* joint_weights;        /* (B,J) or NULL */
  int32_t batch, num_iter;
  float beta_regularizer, beta_regularizer2, kid_regularizer;
  int32_t final_adjust_rots;
  const float* initial_pose_rotvecs; /* (B,3J) or NULL */
  const float* initial_shape_betas;  /* (B,num_initial_betas) or NULL */
  int32_t num_initial_betas;
  const float* initial_kid_factor;   /* (B) or NULL */
  int32_t share_beta;
  int32_t scale_mode;                /* 0; 1 = scale_target, 2 = scale_fit (the last solve, :434-519) */
  float scale_regularizer;
  float* pose_rotvecs;               /* out (B,3J) */
  float* shape_betas;                /* out (B,S) */
  float* trans;                      /* out (B,3) */
  float* kid_factor;                 /* out (B) or NULL */
  float* orientations;               /* out (B,J,3,3) or NULL */
  float* relative_orientations;      /* out (B,J,3,3) or NULL */
  float* scale_corr;                 /* out (B), required with scale_mode != 0 */
  void* workspace;
  size_t workspace_bytes;
  void* hip_stream;
  smplfit_share_allreduce_fn share_allreduce; /* NULL, or the cross-rank sum of a sharded share_beta fit */
  void* share_user;                  /* passed back to share_allreduce */
} smplfit_fit_args;
int smplfit_fit_ex_f32(const smplfit_handle* h, const smplfit_fit_args* args);

/* BodyFitter.fit_with_known_shape (pt/bodyfitter.py:655-838): pose and translation (and, with
 * scale_fit, one scale factor per instance) for KNOWN shape parameters.  num_iter rotation passes
 * against the model posed at the current rotations, then fit_scale_and_translation (:1628-1681) and,
 * with final_adjust_rots, the dependent refinement.
 *   shape_betas (B,num_betas_given) -- betas beyond the handle's count are an error, missing ones are 0
 *   kid_factor (B) or NULL          -- only on a handle created with enable_kid
 *   initial_pose_rotvecs (B,3J) or NULL (rest pose)
 *   target_* / *_weights            -- as smplfit_fit_f32
 * Outputs: pose_rotvecs (B,3J), trans (B,3); scale_corr (B) is written only when scale_fit != 0;
 * orientations / relative_orientations (B,J,3,3) may be NULL.
 * The reference's scale branch multiplies a (B,) scale into (B,3) means and therefore only runs for
 * B == 1 (:1675-1676); here every instance gets its own scale, which is what B == 1 calls return. */
int smplfit_fit_known_shape_f32(const smplfit_handle* h, const float* shape_betas,
                                int num_betas_given, const float* kid_factor,
                                const float* initial_pose_rotvecs, const float* target_vertices,
                                const float* target_joints, const float* vertex_weights,
                                const float* joint_weights, int batch, int num_iter,
                                int final_adjust_rots, int scale_fit, float* pose_rotvecs, float* trans,
                                float* scale_corr, float* orientations, float* relative_orientations,
                                void* workspace, size_t workspace_bytes, void* hip_stream);

/* Stage entry points for parity tests. */

/* First rotation pass of fit (pt/bodyfitter.py:384-394 -> _fit_global_rotations :1321-1416):
 * centres the targets, fits every part's global rotation against the template mesh.
 * glob_rotmats (B,J,3,3) out. */
int smplfit_part_rotations_f32(const smplfit_handle* h, const float* target_vertices,
                               const float* target_joints, const float* vertex_weights,
                               const float* joint_weights, int batch, float* glob_rotmats,
                               void* workspace, size_t workspace_bytes, void* hip_stream);

/* One shape solve (pt/bodyfitter.py:840-1102, _fit_shape -> _fit_shape_gram) for given global
 * rotations on targets that are centred internally exactly as fit() does.  add_mean = 0: trans /
 * vertices / joints stay in the centred frame (stage parity tests); add_mean = 1: the target mean is
 * added back to trans, i.e. fit_with_known_pose (pt/bodyfitter.py:552-653) once the caller has turned
 * pose_rotvecs into global rotations.  vertices_out (B,V,3) / joints_out (B,J,3) may be NULL. */
int smplfit_shape_solve_f32(const smplfit_handle* h, const float* glob_rotmats,
                            const float* target_vertices, const float* target_joints,
                            const float* vertex_weights, const float* joint_weights, int batch,
                            float beta_regularizer, float beta_regularizer2, float kid_regularizer,
                            int add_mean, float* shape_betas, float* trans, float* kid_factor,
                            float* vertices_out, float* joints_out, void* workspace,
                            size_t workspace_bytes, void* hip_stream);

/* The general form of smplfit_shape_solve_f32: fit_with_known_pose (pt/bodyfitter.py:552-653) with the
 * options it hands to the shape solve (:623-638 -> _fit_shape_general :1104-1319).
 *   beta_regularizer_reference (B,num_reference_betas) / kid_regularizer_reference (B) -- the values the
 *   ridge pulls towards (:1224-1255); missing betas are 0; ignored with share_beta, as the all-shared
 *   branch of the reference drops them (pt/lstsq.py:45-47)
 *   share_beta, share_allreduce, share_user -- as smplfit_fit_args
 *   scale_mode, scale_regularizer           -- one more unknown (1 scale_target, 2 scale_fit);
 *   scale_corr (B) out; shape_betas / kid_factor as the reference returns them (undivided); with
 *   add_mean the UNSCALED target mean is added to trans (:642-643).  No mesh outputs with a scale.
 * Zero-initialise the struct; fields left 0 / NULL mean "not given". */
typedef struct smplfit_shape_solve_args {
  const float* glob_rotmats;         /* (B,J,3,3) */
  const float* target_vertices;      /* (B,V,3) */
  const float* target_joints;        /* (B,J,3) or NULL */
  const float* vertex_weights;       /* (B,V) or NULL */
  const float* joint_weights;        /* (B,J) or NULL */
  int32_t batch;
  float beta_regularizer, beta_regularizer2, kid_regularizer;
  int32_t add_mean;
  const float* beta_regularizer_reference; /* (B,num_reference_betas) or NULL */
  int32_t num_reference_betas;
  const float* kid_regularizer_reference;  /* (B) or NULL */
  int32_t share_beta;
  int32_t scale_mode;
  float scale_regularizer;
  float* shape_betas;                /* out (B,S) */
  float* trans;                      /* out (B,3) */
  float* kid_factor;                 /* out (B) or NULL */
  float* scale_corr;                 /* out (B), required with scale_mode != 0 */
  float* vertices_out;               /* out (B,V,3) or NULL */
  float* joints_out;                 /* out (B,J,3) or NULL */
  void* workspace;
  size_t workspace_bytes;
  void* hip_stream;
  smplfit_share_allreduce_fn share_allreduce;
  void* share_user;
} smplfit_shape_solve_args;
int smplfit_shape_solve_ex_f32(const smplfit_handle* h, const smplfit_shape_solve_args* args);

/*
 * Topology transfer — BodyConverter.convert_vertices (pt/bodyconverter.py:128-149): out = M in with the sparse
 * (V_out x V_in) matrix BodyConverter.__init__ loads (pt/bodyconverter.py:31-47; the first V_in columns of the
 * official *_deftrafo_setup.pkl matrix, common.py:425-429).  smplfit_transfer_create copies a CSR matrix (HOST
 * pointers: indptr (V_out + 1), indices / values (nnz)) and uploads it to the current device (flags =
 * SMPLFIT_CREATE_HOST_ONLY: no upload, for tests).  smplfit_transfer_f32: in_vertices (B,V_in,3) ->
 * out_vertices (B,V_out,3), device pointers; the entries of a row are added in CSR order, as the reference's sparse
 * product does.
 */
typedef struct smplfit_transfer smplfit_transfer;
int smplfit_transfer_create(int32_t num_vertices_in, int32_t num_vertices_out, const int32_t* indptr,
                            const int32_t* indices, const float* values, int flags, smplfit_transfer** out);
void smplfit_transfer_destroy(smplfit_transfer* t);
int smplfit_transfer_f32(const smplfit_transfer* t, const float* in_vertices, int batch, float* out_vertices,
                         void* hip_stream);

/*
 * BodyConverter.convert, default branch (pt/bodyconverter.py:49-126: forward of the input model :86, convert_vertices
 * :87, fit of the output model with enable_kid :110-117), as ONE call that keeps every intermediate in the kernels'
 * own instance-innermost layout: the input model's posed vertices never leave their stream buffer, the transfer writes
 * the output model's target stream directly (no (B,V,3) intermediates, no layout pass).
 *   smplfit_convert_plan_create(in, out, transfer, &plan): `in` = handle of body_model_in (no kid unknown), `out` =
 *   handle of body_model_out (normally created with enable_kid, as BodyConverter's fitter is), `transfer` = NULL for
 *   models of one topology (identity).  Returns SMPLFIT_ERR_UNSUPPORTED when one of the models is outside what the
 *   batch-major kernels take — the caller then runs smplfit_forward_f32 + smplfit_transfer_f32 + smplfit_fit_f32.
 *   The plan borrows the two handles (keep them alive) and owns its re-indexed copy of the matrix.
 *   smplfit_convert_f32: inputs pose_rotvecs (B,3 J_in), shape_betas (B,num_betas_given) or NULL, trans (B,3) or NULL
 *   of the INPUT model (its mesh is evaluated without kid_factor, as the reference does); the fit options of
 *   smplfit_fit_f32 (the reference passes beta_regularizer = 0, final_adjust_rots = 0, kid_regularizer = 1e9 or 0);
 *   outputs as smplfit_fit_f32 for the OUTPUT model.  Workspace: smplfit_convert_workspace_bytes(plan, batch).
 */
typedef struct smplfit_convert_plan smplfit_convert_plan;
int smplfit_convert_plan_create(const smplfit_handle* in, const smplfit_handle* out, const smplfit_transfer* transfer,
                                smplfit_convert_plan** plan);
void smplfit_convert_plan_destroy(smplfit_convert_plan* plan);
size_t smplfit_convert_workspace_bytes(const smplfit_convert_plan* plan, int batch);
typedef struct smplfit_convert_args {
  const float* pose_rotvecs;         /* (B,3 J_in) */
  const float* shape_betas;          /* (B,num_betas_given) or NULL */
  int32_t num_betas_given;
  const float* trans;                /* (B,3) or NULL */
  int32_t batch, num_iter;
  float beta_regularizer, beta_regularizer2, kid_regularizer;
  int32_t final_adjust_rots;
  float* out_pose_rotvecs;           /* (B,3 J_out) */
  float* out_shape_betas;            /* (B,S_out) */
  float* out_trans;                  /* (B,3) */
  float* out_kid_factor;             /* (B) or NULL */
  float* out_orientations;           /* (B,J_out,3,3) or NULL */
  float* out_relative_orientations;  /* (B,J_out,3,3) or NULL */
  void* workspace;
  size_t workspace_bytes;
  void* hip_stream;
} smplfit_convert_args;
int smplfit_convert_f32(const smplfit_convert_plan* plan, const smplfit_convert_args* args);

/* Re-reads the SMPLFIT_* tuning variables (INTEGRATION.md lists them).  They are read once, at first use; tests and
 * the A/B tools that switch kernel paths inside one process call this after changing the environment.  Not to be
 * called while other threads are inside the library. */
int smplfit_reload_options(void);

/* Measurement hook (bench.py's roofline leg): launches ONE kernel of the fit `reps` times on
 * `hip_stream` between two HIP events recorded on that same stream and returns the average
 * duration in milliseconds.  The workspace must hold the state a preceding smplfit_fit_f32 call on
 * the same batch left behind.  Synchronises the stream (it is a measurement, not a product call). */
enum smplfit_kernel_id {
  SMPLFIT_KERNEL_POSEDIRS_GEMM = 2, /* K2 v_posed = v_template + pose_feature . posedirs          */
  SMPLFIT_KERNEL_SHAPE_ACCUM = 3,   /* K3 vertex block of the normal equations (batch-major path:
                                       the residual pass k_residual_bm)                            */
  SMPLFIT_KERNEL_SHAPE_SOLVE = 4,   /* K4 fp64 Cholesky solve                                      */
  SMPLFIT_KERNEL_LBS_PARTSUM = 5,   /* K5 vertices at the solution + part sums                     */
  SMPLFIT_KERNEL_PAIR_GRAM = 6,     /* batch-major path: Gramian from the rotations (k_pair_gram_bm) */
  SMPLFIT_KERNEL_TRANSPOSE = 7,     /* batch-major path: targets to the instance-innermost, part-sorted
                                       layout in one pass (k_layout_targets)                        */
  SMPLFIT_KERNEL_TEMPLATE_PARTSUM = 8, /* batch-major path: part sums against the template (first
                                       rotation estimate, k_template_partsum_bm)                    */
  /* the remaining launches of a default fit, so that a caller can attribute the whole step (bench.py's
   * roofline.kernel_ms / fit_breakdown): */
  SMPLFIT_KERNEL_JOINT_STAGE = 9,   /* K1 part rotations + shape prologue (k_joint_stage)            */
  SMPLFIT_KERNEL_REFINE = 10,       /* K6 dependent refinement + epilogue (k_refine_epilogue)       */
  SMPLFIT_KERNEL_GRAM_COMBINE = 11, /* batch-major path: partial sums -> normal-equation record     */
  SMPLFIT_KERNEL_PSUM_COMBINE = 12, /* batch-major path: rows of part sums -> (B, J, 16)            */
  SMPLFIT_KERNEL_JD_TRANSPOSE = 13, /* batch-major path: joint rows instance-innermost              */
  SMPLFIT_KERNEL_MEAN_FINISH = 14,  /* batch-major path: mean of the targets, centred joints        */
  SMPLFIT_KERNEL_LBS_LAST = 15,     /* batch-major path: the last LBS / part-sum pass (adjustable parts only) + combine */
};
int smplfit_time_kernel_f32(const smplfit_handle* h, int kernel_id, int batch, int reps,
                            void* workspace, size_t workspace_bytes, void* hip_stream,
                            float* avg_ms);

/* Test hook: the rotation primitives of the joint-level kernels, evaluated ON THE DEVICE (the
 * arithmetic that ships: hardware rsq / rcp seeds + Newton steps inside proj_SO3), one element per
 * thread, so that the reference's primitive goldens incl. the degenerate inputs (rank 1 / rank 2 /
 * reflection / zero matrices, all four mat2rotvec branches, (anti)parallel vectors) are checked
 * against the device code and not only against its host build.  Device pointers.
 *   PROJ_SO3      a (n,3,3) -> out (n,3,3)   pt/rotation.py:100-110
 *   ROTVEC2MAT    a (n,3)   -> out (n,3,3)   pt/rotation.py:236-258
 *   MAT2ROTVEC    a (n,3,3) -> out (n,3)     pt/rotation.py:261-289
 *   ALIGN_UNIT    a, b (n,3) -> out (n,3,3)  pt/rotation.py:210-224
 *   SWING_TWIST   a = b_ref (n,3), b = [b_tgt | A] (n,12) -> out (n,3,3)  pt/bodyfitter.py:1389-1412 */
enum smplfit_primitive_id {
  SMPLFIT_PRIM_PROJ_SO3 = 0,
  SMPLFIT_PRIM_ROTVEC2MAT = 1,
  SMPLFIT_PRIM_MAT2ROTVEC = 2,
  SMPLFIT_PRIM_ALIGN_UNIT = 3,
  SMPLFIT_PRIM_SWING_TWIST = 4,
};
int smplfit_primitives_f32(int primitive_id, const float* a, const float* b, float* out, int n,
                           void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* SMPLFIT_H_ */
