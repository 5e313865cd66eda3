/*
 * pcpx.h -- C ABI of libpcpx.so: MI355X (gfx950) kNN / radius search / PCA normals.
 *
 * This is the drop-in boundary for the one data-parallel hot path of
 * Q-Minh/point-cloud-processing (paths below are relative to the reference root):
 *
 *   pcp::basic_linked_octree_t ctor / nearest_neighbours / range_search
 *       include/pcp/octree/linked_octree.hpp:83-121, :245-254, :264-276
 *   pcp::basic_linked_kdtree_t ctor / nearest_neighbours / range_search
 *       include/pcp/kdtree/linked_kdtree.hpp:100-135, :200-263, :270-277
 *   pcp::estimate_normal / pcp::algorithm::estimate_normals
 *       include/pcp/common/normals/normal_estimation.hpp:32-78
 *       include/pcp/algorithm/estimate_normals.hpp:50-93, :116-164
 *
 * The reference is a header-only C++17 template library with no FFI; the C++17 mirror headers
 * under include/pcp/ (same names, namespaces and signatures) call these entry points, and
 * INTEGRATION.md shows the binding a maintainer of the reference would add.  Everything here is
 * POD: plain pointers and sizes, no C++ or torch types.
 *
 * Conventions
 *   - Points are float32 xyz, array-of-structures (pcp::point_t layout,
 *     include/pcp/common/points/point.hpp:21-93), n < 2^32 - 1.
 *   - Results are INDICES INTO THE INPUT ARRAY (the host wrappers map index -> Element).
 *   - kNN rows are ascending in (squared distance, index); a row holds `count[q] <= k` valid
 *     entries followed by 0xFFFFFFFF padding.  Points inside the eps-box around the query
 *     (|dx|<eps && |dy|<eps && |dz|<eps, include/pcp/common/vector3d_queries.hpp:47-64) are
 *     excluded, exactly like linked_octree_node.hpp:540 and linked_kdtree.hpp:461-475.
 *   - Squared distances are computed as the reference does (include/pcp/common/norm.hpp:102-112):
 *     d = p - q per axis, dx*dx + dy*dy + dz*dz left to right in float32, NO fused multiply-add.
 *   - Sphere range search returns every point with d2 <= r*r (include/pcp/common/sphere.hpp:27-35),
 *     the query point included, in unspecified order.
 *   - Functions return PCPX_OK (0) or a negative pcpx_status; pcpx_last_error() gives the text of
 *     the calling thread's last failure.  Nothing throws across this boundary.
 *   - `*_dev` variants take DEVICE pointers and a hipStream_t (as void*); they enqueue work on that
 *     stream and return without synchronising.  A NULL stream is the legacy default stream (ordered against
 *     the caller's other default-stream work), never a private one.  The others take HOST pointers and are
 *     synchronous; they run on a stream of their own.  Every entry point restores the calling thread's
 *     current device before it returns.
 *   - There is no CPU fallback: without a usable HIP device every compute entry point fails
 *     with PCPX_ERR_DEVICE.
 */
#ifndef PCPX_H
#define PCPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCPX_ABI_VERSION 5

typedef enum pcpx_status {
    PCPX_OK = 0,
    PCPX_ERR_INVALID = -1,  /* bad argument */
    PCPX_ERR_DEVICE = -2,   /* HIP error / no device */
    PCPX_ERR_ALLOC = -3,    /* out of host or device memory */
    PCPX_ERR_CAPACITY = -4, /* caller buffer too small; required size reported */
    PCPX_ERR_UNSUPPORTED = -5
} pcpx_status;

typedef struct pcpx_index pcpx_index; /* opaque: device buffers + stream */

/* Build flags */
#define PCPX_BUILD_USE_GRID 1u /* voxel_grid given: points outside it are silently dropped         \
                                  (linked_octree_node.hpp:174-175, linked_octree.hpp:83-91);        \
                                  also the curve-key quantisation box (e.g. the union of per-rank      \
                                  boxes after the RCCL all-gather).  Otherwise the tight bounding   \
                                  box of the input is used (linked_octree.hpp:103-121).           */

#define PCPX_BUILD_COARSE_ORDER 2u /* for an index that is rebuilt after a query pass or two (streaming clouds): points are     \
                                     sorted along the curve only as finely as the size of their top-level bucket asks for    \
                                     (one radix pass fewer on a uniform cloud).  Query RESULTS are unaffected -- the tree's  \
                                     boxes come from the points whatever their order; on strongly clustered clouds leaves    \
                                     are a little less compact (k-NN 3 % slower at 10 M clustered points, the rebuild 5 %    \
                                     faster at 50 M uniform).  Additive: the reference has no counterpart.                 */

#define PCPX_BUILD_SHARD 4u /* RANK-LOCAL index for the multi-GPU path (one process per GPU, the cloud replicated): the     \
                              handle indexes only what shard `shard_rank` of `shard_world` of the curve-sorted queries      \
                              (pcpx_shard_range over pcpx_index_size) can reach -- the cells of the curve that hold the     \
                              shard and a halo of cells around them -- instead of the whole cloud: curve keys for every     \
                              point, then sort + leaves + boxes for about 1 / world of them.  The *_self_dev entry points    \
                              then take slices INSIDE that shard (positions of the WHOLE cloud's order, as before) and       \
                              return exactly what the whole-cloud index returns: every query's k-th distance is checked      \
                              against the cells the handle holds, and a query whose search ball leaves them is answered      \
                              again after the handle has taken the missing cells in (it keeps them: a static index pays     \
                              once).  Because of that check these calls synchronise the stream before they return -- except  \
                              when the same question (k, eps, slice) was checked on this tree before: then they only enqueue.\
                              pcpx_index_size / _bbox describe the whole cloud.  Entry points that need the whole cloud      \
                              indexed (arbitrary query batches, range lists, host-pointer forms) fail with                   \
                              PCPX_ERR_UNSUPPORTED on such a handle.  No collective is involved: ranks agree on the grid     \
                              beforehand (PCPX_BUILD_USE_GRID, pcpx_comm_global_grid_dev).  Additive: the reference is        \
                              single-process.                                                                             */
#define PCPX_BUILD_BORROW_CLOUD 8u /* *_dev builds only: the handle reads the caller's d_xyz in place instead of copying it    \
                                      (12 B/point less to write per rebuild); the array must stay valid and unchanged until   \
                                      the handle is rebuilt or destroyed.  Without the flag the handle copies, like the       \
                                      reference's containers do (linked_kdtree.hpp:107).                                   */

#define PCPX_BUILD_SHARD_RANGE 16u /* with PCPX_BUILD_SHARD: the shard is curve positions [shard_first, shard_first + shard_count) of the   \
                                      whole cloud's order (shard_first a multiple of 64; clipped to the cloud) instead of                 \
                                      pcpx_shard_range(shard_rank, shard_world) -- the cut of pcpx_shard_cuts_by_cost, which gives        \
                                      every rank the same WORK rather than the same number of queries.                                 */

typedef struct pcpx_build_params {
    uint32_t struct_size; /* = sizeof(pcpx_build_params) (callers built against ABI 3 / 4 pass its first 32 / 48 bytes: accepted) */
    uint32_t flags;
    float grid_min[3];
    float grid_max[3];
    /* PCPX_BUILD_SHARD */
    uint32_t shard_rank;
    uint32_t shard_world;
    uint32_t shard_k_hint; /* the k the shard will be asked for: sizes the halo (0 = 32).  Only a performance hint. */
    uint32_t reserved;     /* 0 */
    /* PCPX_BUILD_SHARD_RANGE */
    uint64_t shard_first;
    uint64_t shard_count;
} pcpx_build_params;

int pcpx_abi_version(void);
const char* pcpx_last_error(void);
int pcpx_device_count(int* out_count);

/* ---- index construction: replaces the octree / kd-tree constructors ---------------------- */
int pcpx_index_create(const float* xyz, uint64_t n, const pcpx_build_params* params, int device,
                      pcpx_index** out);
int pcpx_index_create_dev(const float* d_xyz, uint64_t n, const pcpx_build_params* params, int device,
                          void* stream, pcpx_index** out);
/* Re-index a new cloud in place, reusing device buffers (BASELINE config 5: rebuild per iteration). */
int pcpx_index_rebuild(pcpx_index* idx, const float* xyz, uint64_t n, const pcpx_build_params* params);
int pcpx_index_rebuild_dev(pcpx_index* idx, const float* d_xyz, uint64_t n, const pcpx_build_params* params);
void pcpx_index_destroy(pcpx_index* idx);
/* size(): number of points inserted (linked_octree.hpp:127) -- input points minus out-of-grid ones. */
int pcpx_index_size(pcpx_index* idx, uint64_t* out_n);
/* voxel_grid() / aabb(): {minx,miny,minz,maxx,maxy,maxz} (linked_octree.hpp:144, linked_kdtree.hpp:187). */
int pcpx_index_bbox(pcpx_index* idx, float out6[6]);
/* A rank-local handle (PCPX_BUILD_SHARD) described: out[0] points in the local tree, [1] first global curve position of its
 * core, [2] points of the core, [3] / [4] the shard (first position, count), [5] halo width in cells of the selection grid
 * (64 per axis), [6] queries the last coverage check sent round again, [7] times the handle has taken more cells in.
 * PCPX_ERR_INVALID for a whole-cloud handle. */
int pcpx_index_shard_info(pcpx_index* idx, uint64_t out[8]);
/* pcp::bounding_box over a host slice (axis_aligned_bounding_box.hpp:214-251): the per-rank step
 * before the bounding-box all-gather of the multi-GPU path. */
int pcpx_bounding_box(const float* xyz, uint64_t n, int device, float out6[6]);
int pcpx_bounding_box_dev(const float* d_xyz, uint64_t n, int device, void* stream, float* d_out6);

/* ---- k nearest neighbours: replaces nearest_neighbours ----------------------------------- */
/* Every indexed point queries the cloud (the estimate_normals / kNN-graph shape).  Row i of the
 * outputs belongs to input point i.  out_d2 may be NULL. */
int pcpx_knn_self(pcpx_index* idx, uint32_t k, float eps, uint32_t* out_idx, uint32_t* out_count,
                  float* out_d2);
/* Arbitrary query points. */
int pcpx_knn_batch(pcpx_index* idx, const float* q_xyz, uint64_t nq, uint32_t k, float eps,
                   uint32_t* out_idx, uint32_t* out_count, float* out_d2);
/* Device-pointer form.  Only curve-sorted positions [sorted_first, sorted_first+sorted_count) are
 * processed (rows of the other points are left untouched) -- the per-rank query shard of the
 * multi-GPU path; pass 0, UINT64_MAX for all.  sorted_first must be a multiple of 64. */
int pcpx_knn_self_dev(pcpx_index* idx, uint32_t k, float eps, uint64_t sorted_first, uint64_t sorted_count,
                      uint32_t* d_out_idx, uint32_t* d_out_count, float* d_out_d2);
int pcpx_knn_batch_dev(pcpx_index* idx, const float* d_q_xyz, uint64_t nq, uint32_t k, float eps,
                       uint32_t* d_out_idx, uint32_t* d_out_count, float* d_out_d2);
/* pcpx_knn_self_dev with `row_stride` entries between the rows of d_out_idx / d_out_d2 (0: k; otherwise >= k, k <= 32; entries
 * k .. row_stride of a row are 0xFFFFFFFF / +inf).  The reference returns one std::vector per query
 * (include/pcp/octree/linked_octree.hpp:245-254), so the row pitch is this library's to choose: with row_stride = 16 (k = 15 or 16;
 * 8 for k = 7, 8; 32 for k = 31, 32) and outputs aligned to 16 bytes a row is ONE aligned 64-byte piece, written with 16-byte
 * stores.  Packed 60-byte rows scattered by input index cost every partial 32-byte sector twice at the memory side (read for
 * ownership + write back): 1.57 GB written per 10 M queries for 0.76 GB of payload (profiles/r04_hbm_traffic.json). */
int pcpx_knn_self_strided_dev(pcpx_index* idx, uint32_t k, float eps, uint64_t sorted_first, uint64_t sorted_count,
                              uint32_t row_stride, uint32_t* d_out_idx, uint32_t* d_out_count, float* d_out_d2);
/* Rows by CURVE POSITION, device resident (additive): row p of every output belongs to the p-th point of the curve order,
 * i.e. to input point perm[p] (pcpx_index_perm_dev); neighbour indices inside the rows are input indices as everywhere.
 * A wavefront's 64 rows are then one contiguous piece of each output (the input-order form scatters 60-byte rows over the
 * whole array).  Any of d_opt_normals / d_opt_d2 may be NULL; slice arguments as pcpx_knn_self_dev. */
int pcpx_knn_self_curve_order_dev(pcpx_index* idx, uint32_t k, float eps, uint64_t sorted_first, uint64_t sorted_count,
                                  uint32_t* d_out_idx, uint32_t* d_out_count, float* d_opt_d2, float* d_opt_normals);
/* The curve order itself: d_out_perm[p] = input index of the p-th point (pcpx_index_size entries; a rank-local handle: the
 * entries of its core only, at their global positions), d_opt_out_position_of[i] = position of input point i (n entries,
 * 0xFFFFFFFF for a point that is not indexed).  Either may be NULL.  Enqueued on the handle's stream. */
int pcpx_index_perm_dev(pcpx_index* idx, uint32_t* d_out_perm, uint32_t* d_opt_out_position_of);

/* ---- radius search: replaces range_search ------------------------------------------------ */
/* Count only (what examples/filter_point_cloud_noise_by_density.cpp:81-90 consumes). */
int pcpx_range_count_self(pcpx_index* idx, float radius, uint32_t* out_count);
int pcpx_range_count_batch(pcpx_index* idx, const float* q_xyz, uint64_t nq, float radius, uint32_t* out_count);
int pcpx_range_count_self_dev(pcpx_index* idx, float radius, uint64_t sorted_first, uint64_t sorted_count,
                              uint32_t* d_out_count);
/* The same with the count of the p-th point of the curve order at d_out_count[p] (positions of the whole cloud's order; rows of
 * a slice are one contiguous piece, written 256 bytes per query group; pcpx_index_perm_dev gives the order).  Additive: the
 * reference returns results per element (include/pcp/octree/linked_octree.hpp:264-276); a consumer that reduces the counts
 * (examples/filter_point_cloud_noise_by_density.cpp:81-90 thresholds them) does not care about their order. */
int pcpx_range_count_self_curve_order_dev(pcpx_index* idx, float radius, uint64_t sorted_first, uint64_t sorted_count,
                                          uint32_t* d_out_count);
/* range_search around EVERY indexed point, lists, device resident (additive; the reference returns one std::vector per call,
 * include/pcp/octree/linked_octree.hpp:264-276): d_out_offsets has n + 1 entries (n = input points; list of input point i =
 * d_out_idx[offsets[i] .. offsets[i + 1]), a point outside the voxel grid has none), each list in tree order.  Counts, a 64-bit
 * exclusive scan and the fill run on the handle's stream; the call waits once, for the total, which it returns in *out_total.
 * PCPX_ERR_CAPACITY (offsets filled, *out_total set) if idx_capacity indices do not hold them or d_out_idx is NULL: allocate and
 * call again.  Enqueues the fill and returns. */
int pcpx_range_lists_self_dev(pcpx_index* idx, float radius, uint64_t* d_out_offsets, uint32_t* d_out_idx, uint64_t idx_capacity,
                              uint64_t* out_total);
/* Lists, CSR: out_offsets has nq+1 entries; out_idx receives offsets[nq] indices.  If idx_capacity is
 * too small (or out_idx is NULL) the offsets are still filled and PCPX_ERR_CAPACITY is returned, so
 * the caller can allocate offsets[nq] entries and call again. */
int pcpx_range_sphere_batch(pcpx_index* idx, const float* q_xyz, const float* radii, float radius, uint64_t nq,
                            uint64_t* out_offsets, uint32_t* out_idx, uint64_t idx_capacity);
/* Axis-aligned box ranges, 6 floats each {min,max}, inclusive on both ends
 * (axis_aligned_bounding_box.hpp:111-125). */
int pcpx_range_aabb_batch(pcpx_index* idx, const float* boxes6, uint64_t nb, uint64_t* out_offsets,
                          uint32_t* out_idx, uint64_t idx_capacity);

/* ---- PCA normals: replaces estimate_normal / estimate_normals ---------------------------- */
/* estimate_normals with knn_map = k nearest neighbours of each indexed point.  out_normals is n x 3
 * (pcp::normal_t layout); opt_out_idx (n x k) / opt_out_count (n) may be NULL.  Sign is arbitrary,
 * as in the reference (test/algorithm/estimate_normals.cpp:58-59). */
int pcpx_normals_knn_self(pcpx_index* idx, uint32_t k, float eps, float* out_normals, uint32_t* opt_out_idx,
                          uint32_t* opt_out_count);
/* The same with the rows in CURVE ORDER (additive; no reference counterpart): row p belongs to the p-th inserted point of
 * the index's sorted order, i.e. to input point opt_out_perm[p] (pcpx_index_size rows); neighbour indices inside the rows
 * are input indices as everywhere.  opt_out_position_of (n entries, 0xFFFFFFFF for a point outside the voxel grid) is the
 * inverse: the row of input point i.  The library computes the rows slice by slice along the curve and copies a finished
 * slice to the host while the next ones are computed, which the input-order form cannot do (its rows are scattered over
 * the whole output): use this form when the rows are consumed through a table anyway. */
int pcpx_normals_knn_self_curve_order(pcpx_index* idx, uint32_t k, float eps, float* opt_out_normals, uint32_t* out_idx,
                                      uint32_t* out_count, uint32_t* opt_out_perm, uint32_t* opt_out_position_of);
int pcpx_normals_knn_self_dev(pcpx_index* idx, uint32_t k, float eps, uint64_t sorted_first,
                              uint64_t sorted_count, float* d_out_normals, uint32_t* d_opt_out_idx,
                              uint32_t* d_opt_out_count);
/* The same with a row pitch (see pcpx_knn_self_strided_dev). */
int pcpx_normals_knn_self_strided_dev(pcpx_index* idx, uint32_t k, float eps, uint64_t sorted_first, uint64_t sorted_count,
                                      uint32_t row_stride, float* d_out_normals, uint32_t* d_opt_out_idx,
                                      uint32_t* d_opt_out_count);
/* ---- tangent planes / mean neighbour distance (the callers right next to the normal loop) -------- */
/* estimate_tangent_planes (include/pcp/algorithm/estimate_tangent_planes.hpp:50-98): plane of point i =
 * (center_of_geometry of its k nearest neighbours, include/pcp/common/vector3d_queries.hpp:77-99; their PCA
 * normal).  out_centroids and out_normals are n x 3. */
int pcpx_tangent_planes_knn_self(pcpx_index* idx, uint32_t k, float eps, float* out_centroids, float* out_normals);
/* average_distances_to_neighbors (include/pcp/algorithm/average_distance_to_neighbors.hpp:32-73): for every
 * indexed point the mean Euclidean distance to its k nearest neighbours (NaN for an empty neighbourhood). */
int pcpx_mean_knn_distance_self(pcpx_index* idx, uint32_t k, float eps, float* out_mean_dist);
/* Device form of both plus normals; any output may be NULL (not all). Sorted-slice arguments as pcpx_knn_self_dev. */
int pcpx_neighbourhoods_self_dev(pcpx_index* idx, uint32_t k, float eps, uint64_t sorted_first, uint64_t sorted_count,
                                 float* d_opt_normals, float* d_opt_centroids, float* d_opt_mean_dist);

/* ---- normal orientation (the step right after the normal loop) ------------------------------------ */
/* propagate_normal_orientations (include/pcp/algorithm/estimate_normals.hpp:187-302) over the kNN graph the
 * query kernels produce: vertex i -> knn_idx[i*k .. i*k+count[i]) in row order (opt_knn_count NULL = k each).
 * As in the reference the root is the first point of largest z, its normal becomes (0,0,1), and a
 * breadth-first search (include/pcp/graph/search.hpp:36-85) flips the normal of every newly reached vertex v
 * iff dot(n(v), n(parent)) < 0 and not |dot| < 1e-5.  The visit order decides which parent a vertex gets, so
 * this is a sequential host pass over the rows (no GPU work; xyz, rows and normals are host arrays);
 * normals (n x 3) are updated in place.  opt_out_reached (may be NULL) receives the number of vertices
 * reached from the root.  PCPX_ERR_INVALID if a row holds an index >= n. */
int pcpx_propagate_normal_orientations(const float* xyz, uint64_t n, const uint32_t* knn_idx,
                                       const uint32_t* opt_knn_count, uint32_t k, float* normals,
                                       uint64_t* opt_out_reached);

/* The same on the device, with the same result: a level-synchronous search in which every newly reached vertex
 * is claimed by its earliest (frontier position, edge index) -- the order the reference's queue would have
 * produced -- so the flips are bit-identical with the host form.  All arrays are device arrays; synchronises
 * `stream` once per 8 BFS levels.  opt_out_levels (may be NULL) receives the depth of the search. */
int pcpx_propagate_normal_orientations_dev(const float* d_xyz, uint64_t n, const uint32_t* d_knn_idx,
                                           const uint32_t* d_opt_knn_count, uint32_t k, float* d_normals, int device,
                                           void* stream, uint64_t* opt_out_reached, uint32_t* opt_out_levels);
/* estimate_normals + propagate_normal_orientations of every indexed point in one call (examples/
 * normals_estimation.cpp:86-117): fused kNN + PCA normals, then the device orientation pass over the rows, which
 * never leave the GPU unless asked for (opt_out_idx n x k, opt_out_count n).  PCPX_ERR_UNSUPPORTED if points were
 * dropped by an explicit voxel grid (they have no neighbourhood). */
int pcpx_oriented_normals_knn_self(pcpx_index* idx, uint32_t k, float eps, float* out_normals, uint32_t* opt_out_idx,
                                   uint32_t* opt_out_count, uint64_t* opt_out_reached);

/* The orientation pass alone for normals the caller already has (host array n x 3, input order, updated in place):
 * the k nearest neighbours of every indexed point are computed on the GPU and never copied out. */
int pcpx_orient_normals_knn_self(pcpx_index* idx, uint32_t k, float eps, float* normals, uint64_t* opt_out_reached);

/* estimate_normal over explicit neighbourhoods: row q = nbr_idx[q*k .. q*k+count[q]) indexes the
 * index's points.  opt_out_evals (nq x 3, ascending eigenvalues) may be NULL. */
int pcpx_normals_from_knn(pcpx_index* idx, const uint32_t* nbr_idx, const uint32_t* count, uint64_t nq,
                          uint32_t k, float* out_normals, float* opt_out_evals);
/* estimate_normal over an arbitrary point set (no index needed): xyz is m x 3. */
int pcpx_estimate_normal(const float* xyz, uint64_t m, int device, float out_normal[3]);
/* The same for many point sets in one launch -- what algorithm::estimate_normals does with an arbitrary (user) knn_map:
 * neighbourhood r = points [offsets[r], offsets[r+1]) of xyz (CSR, nrows + 1 offsets); out_normals is nrows x 3. */
int pcpx_estimate_normals_batch(const float* xyz, const uint64_t* offsets, uint64_t nrows, int device,
                                float* out_normals);

/* ---- consumers of sphere ranges: bilateral filter and WLOP ------------------------------------------- */
/* The reference's kd-tree sphere range has two in-library consumers; both loop over every point's range and
 * reduce it to a few numbers, so here the loop is fused into the range walk (no neighbour list is written).
 * Arithmetic is the reference's, in float; what is not specified by its interface -- the order in which a
 * range's points are summed -- is tree order here: results agree with the reference to float rounding
 * (tests/test_gpu_filters.py states the tolerances), not bit for bit.
 *
 * bilateral_filter_points (include/pcp/algorithm/bilateral_filter.hpp:303-428): `iterations` (params_t::K)
 * rounds of p' = sum w * projection(p) / sum w over the range of radius 2 * sigmaf around p, w = gaussian(sigmaf,
 * |s - p|) * gaussian(sigmag, |projection - s|), projection onto the neighbour's tangent plane (:47-101); the
 * range tree is rebuilt over the moved points every round, the normals stay.  xyz, normals, out_xyz: n x 3.
 * bilateral_filter_normals (:460-574): the normals move instead (n' = normalize(J n), J the Jacobian of the
 * filter at the point, :103-269), the points and their tree stay.
 * sigmaf / sigmag are converted to float like the reference converts params_t's doubles to the point's scalar
 * type.  iterations == 0 copies the input.  out may alias the array it replaces.  A point with a NaN
 * coordinate has an empty range and comes back NaN.  The *_dev forms take device arrays, run on `stream`
 * (NULL = the legacy default stream) and return after the work has completed. */
int pcpx_bilateral_filter_points(const float* xyz, const float* normals, uint64_t n, double sigmaf, double sigmag,
                                 uint64_t iterations, int device, float* out_xyz);
int pcpx_bilateral_filter_normals(const float* xyz, const float* normals, uint64_t n, double sigmaf, double sigmag,
                                  uint64_t iterations, int device, float* out_normals);
int pcpx_bilateral_filter_points_dev(const float* d_xyz, const float* d_normals, uint64_t n, double sigmaf,
                                     double sigmag, uint64_t iterations, int device, void* stream, float* d_out_xyz);
int pcpx_bilateral_filter_normals_dev(const float* d_xyz, const float* d_normals, uint64_t n, double sigmaf,
                                      double sigmag, uint64_t iterations, int device, void* stream,
                                      float* d_out_normals);
/* wlop::wlop (include/pcp/algorithm/wlop.hpp:287-428): weighted locally optimal projection of n_samples points
 * onto the cloud.  The reference seeds x with a std::random_device shuffle of the cloud (:331-343); here the
 * caller names the seed points (sample[i] < n, n_samples <= n), everything after that is the reference's:
 * v_j over the cloud (uniform != 0, :29-65), then per iteration a tree over x, w_i (:67-105), and
 * x' = median of the cloud around x (:107-170) + mu-weighted repulsion among x (:172-229), radius h.
 * out_xyz is n_samples x 3 (iterations == 0: the seed points). */
int pcpx_wlop(const float* xyz, uint64_t n, const uint64_t* sample, uint64_t n_samples, double mu, double h,
              uint64_t iterations, int uniform, int device, float* out_xyz);
int pcpx_wlop_dev(const float* d_xyz, uint64_t n, const uint64_t* d_sample, uint64_t n_samples, double mu, double h,
                  uint64_t iterations, int uniform, int device, void* stream, float* d_out_xyz);

/* ---- device memory for callers without a HIP toolchain --------------------------------------------- */
/* The *_dev forms take raw device pointers; a C or C++ host that does not compile against HIP gets them here
 * (hipMalloc / hipFree / hipMemcpyAsync + stream synchronisation behind the ABI).  include/pcp/gpu/device_index.hpp builds its
 * device-resident result handle (pcp::gpu::device_rows_t) on these. */
int pcpx_device_malloc(uint64_t bytes, int device, void** out_ptr);
void pcpx_device_free(void* d_ptr, int device);
int pcpx_device_upload(void* d_dst, const void* src, uint64_t bytes, int device, void* stream);   /* synchronous */
int pcpx_device_download(void* dst, const void* d_src, uint64_t bytes, int device, void* stream); /* synchronous */
/* Destroying an index keeps its device blocks for the next index of the same device (a drop-in caller constructs containers in a
 * loop, benchmark/spatial_data_structures_benchmark.cpp:108-148, and hipMalloc / hipFree cost more than a small build): at most
 * PCPX_DEVICE_CACHE_MB (environment; default 2048, 0 = keep nothing) of idle blocks per device, and beyond 256 MB never more
 * than a quarter of the device memory that is free at that moment.  A process that shares the GPU with another allocator (PyTorch
 * ...) calls this after it has destroyed its indexes: it gives the idle blocks back. */
int pcpx_device_trim(int device);

/* ---- multi-GPU: one process per GPU ------------------------------------------------------------ */
/* Contiguous, 64-aligned shard of the curve-sorted query order for `rank` of `world`. */
int pcpx_shard_range(uint64_t n, uint32_t rank, uint32_t world, uint64_t* out_first, uint64_t* out_count);

/* Shards of equal WORK.  Equal numbers of queries are not equal work on a clustered cloud (BASELINE configs[3]: the slowest of
 * eight equal-count shards takes 14 % longer than their mean).  pcpx_knn_group_costs_dev runs the k-NN walk (no rows are written)
 * for one query group in every `group_stride` of the whole cloud's curve order -- group i * stride + stride / 2 stands for groups
 * [i * stride, (i + 1) * stride) -- on a WHOLE-CLOUD handle and leaves four event counts per sampled group in d_out_events
 * ({node expansions, leaves looked at lane-per-query [bits 24-31: walk rounds after the first], leaves looked at in the packed form
 * [bits 20-31: lanes those rounds ran for], folds << 16 | packed steps}):
 * integers that depend on the cloud, the grid and (k, eps) alone, so every rank of a job computes the same table from its
 * replica of the cloud, and the all-gather of the boxes stays the only collective.  *out_samples = groups / group_stride
 * (PCPX_ERR_CAPACITY if `capacity` samples do not hold them; d_out_events may be NULL to ask).  1 <= k <= 32.
 * pcpx_shard_cuts_by_cost (host arrays) turns the table into world + 1 positions: rank r answers
 * [out_first[r], out_first[r + 1]) -- multiples of 64 -- e.g. through PCPX_BUILD_SHARD_RANGE. */
int pcpx_knn_group_costs_dev(pcpx_index* idx, uint32_t k, float eps, uint32_t group_stride, uint32_t* d_out_events,
                             uint64_t capacity, uint64_t* out_samples);
int pcpx_shard_cuts_by_cost(uint64_t n, uint32_t world, uint32_t group_stride, const uint32_t* events, uint64_t nsamples,
                            uint64_t* out_first);

/* The path's one collective -- an RCCL all-gather of the per-rank bounding boxes (6 floats = 24 B per rank over
 * xGMI) -- behind the ABI, so that a C++ host of the drop-in headers has the multi-GPU path too.  The reference has
 * no counterpart (it is single-process).  Sequence per rank: communicator (either created here from an id that rank 0
 * generates and shares out of band -- MPI_Bcast, a file, a socket -- or an existing ncclComm_t of the host program);
 * pcpx_comm_global_grid_dev over this rank's slice of the input -> the same grid on every rank; pcpx_index_create_dev
 * of the WHOLE cloud with PCPX_BUILD_USE_GRID on that grid; *_self_dev calls restricted to pcpx_shard_range.
 * librccl.so.1 is loaded at the first pcpx_comm_* call; PCPX_ERR_UNSUPPORTED if it cannot be. */
typedef struct pcpx_comm pcpx_comm;
#define PCPX_COMM_ID_BYTES 128
int pcpx_comm_unique_id(char out_id[PCPX_COMM_ID_BYTES]);
int pcpx_comm_init_rank(const char id[PCPX_COMM_ID_BYTES], int world, int rank, int device, pcpx_comm** out);
/* adopt an ncclComm_t the host program already has (it stays the caller's: pcpx_comm_destroy does not destroy it) */
int pcpx_comm_wrap(void* nccl_comm, int world, int rank, int device, pcpx_comm** out);
void pcpx_comm_destroy(pcpx_comm* comm);
/* all-gather of one box per rank: d_local6 (6 floats) -> d_all (world x 6 floats), enqueued on `stream` */
int pcpx_comm_allgather_boxes_dev(pcpx_comm* comm, const float* d_local6, float* d_all, void* stream);
/* bounding box of this rank's device-resident slice, all-gather, union: the common grid, returned on the host
 * (synchronises `stream`) */
int pcpx_comm_global_grid_dev(pcpx_comm* comm, const float* d_xyz_slice, uint64_t n_slice, void* stream,
                              float out_grid6[6]);

/* The host-pointer entry points stage through device buffers that stay with the handle between calls (no
 * hipMalloc / hipFree per call); this releases the ones not in use.  pcpx_index_destroy releases everything. */
int pcpx_index_trim(pcpx_index* idx);
/* Block until everything enqueued on the index's stream has finished. */
int pcpx_index_synchronize(pcpx_index* idx);

/* ---- measurement --------------------------------------------------------------------------- */
/* Per-kernel device time, measured with hipEvents recorded on the index's stream around every kernel
 * family launched between pcpx_profile_begin and pcpx_profile_end (which synchronises the stream). */
typedef enum pcpx_kernel_family {
    PCPX_K_BUILD = 0,   /* bbox + curve keys + sort + leaves + boxes (one interval per build) */
    PCPX_K_KNN = 1,     /* k_knn */
    PCPX_K_NORMALS = 2, /* k_normals */
    PCPX_K_RANGE = 3,   /* k_range / k_range_aabb */
    PCPX_K_QUERY_PREP = 4, /* query curve sort + seeds of *_batch calls */
    PCPX_K_FAMILIES = 5
} pcpx_kernel_family;
typedef struct pcpx_profile {
    uint32_t launches[PCPX_K_FAMILIES];
    float total_ms[PCPX_K_FAMILIES];
} pcpx_profile;
/* Diagnostic build of the self-kNN kernel (k <= 16): out_stats = {leaves visited, node expansions,
 * compactions, keys appended, query groups, seed leaves, 0...} summed over the launch in out_stats[0..16),
 * followed (capacity permitting) by {start, end (100 MHz ticks), groups done, slowest group ticks, its id}
 * of every persistent wave. */
int pcpx_debug_knn_stats(pcpx_index* idx, uint32_t k, float eps, uint64_t* out_stats, uint64_t capacity);
/* Where the throughput kNN kernel (k <= 32) applies the reference's eps-box exclusion
 * (include/pcp/common/vector3d_queries.hpp:47-64): 0 = chosen per launch from eps and the cloud's point spacing (the
 * default), 1 = on buffered keys when they are folded into the best-list, 2 = on every candidate.  Results are the same;
 * the tests run all three. */
int pcpx_debug_eps_test_mode(pcpx_index* idx, int mode);
/* Switches of the handle that change how work is done, never what comes out (the tests run both sides of each):
 * "long_groups_first" (default 1): a self-kNN launch that repeats the previous one's question on the same tree hands its query
 * groups out by the times that launch recorded, the longest first; "gather_outputs" (default 0: measured slower): input-order
 * normals and counts are written at curve positions and permuted by a gather instead of being scattered from the search kernel;
 * "gather_counts" (default 0: measured slower as well): the same for pcpx_range_count_self_dev's counts. */
int pcpx_debug_set(pcpx_index* idx, const char* name, int64_t value);
/* Figures of the handle: "build_redos" (builds repeated because the leaf kernel met a run of sort words it could not order),
 * "full_buckets" (top-digit buckets that take every radix pass since), "schedule_state" (0: nothing recorded, 1: group times
 * recorded, 2: the recorded order is in use). */
int pcpx_debug_get(pcpx_index* idx, const char* name, int64_t* out_value);
/* What the handle's last recorded self-kNN launch took per query group (shader-clock ticks / 64, search only; host array of
 * *out_groups entries, the launch's groups in curve order; 0 groups: nothing recorded).  PCPX_ERR_CAPACITY reports the size. */
int pcpx_debug_group_times(pcpx_index* idx, uint32_t* out_ticks, uint64_t capacity, uint64_t* out_groups);
/* Diagnostic access to the build's radix sort: stable sort of 64-bit words by their bits [first_bit, 64)
 * (first_bit a multiple of 8): words that agree on those bits keep their input order. */
int pcpx_debug_sort_keys(const uint64_t* keys, uint64_t n, int first_bit, int device, uint64_t* out_keys);
int pcpx_profile_begin(pcpx_index* idx);
int pcpx_profile_end(pcpx_index* idx, pcpx_profile* out);

/* ---- kd-tree for K > 3 coordinates -------------------------------------------------------------------------------------
 * The reference's pcp::basic_linked_kdtree_t is generic in K (include/pcp/kdtree/linked_kdtree.hpp:64); the index above holds
 * three coordinates (K = 1, 2, 3 go through it, missing axes carried as 0).  For 4 <= K <= PCPX_KD_MAX_DIMS the two queries of the
 * class are answered by exhaustive search on the GPU (csrc/pcpx_kd.hip), with the reference's results:
 *   pcpx_kd_knn_batch        nearest_neighbours (linked_kdtree.hpp:200-262, :436-540): per query the k points of smallest squared
 *                            distance (sum over the axes in order, common/norm.hpp:123-141), nearest first, a point with
 *                            |p[a] - q[a]| < eps on every axis left out; ties in index order.  out_idx nq x k (0xFFFFFFFF beyond
 *                            out_count[q]), opt_out_d2 nq x k (+inf beyond) or NULL.  Any k.
 *   pcpx_kd_range_aabb_batch range_search with a kd box (linked_kdtree.hpp:270-311): boxes nb x 2 dims (min[dims] then max[dims]),
 *                            min <= p <= max on every axis; CSR offsets nb + 1 and indices, unordered inside a box;
 *                            PCPX_ERR_CAPACITY with the offsets filled if idx_capacity (or a NULL out_idx) does not hold them.
 * points / queries: host arrays, row-major (n x dims).  Fewer than 2^32 - 1 points.  One call at a time per handle. */
#define PCPX_KD_MAX_DIMS 16
typedef struct pcpx_kd_index pcpx_kd_index;
int pcpx_kd_create(const float* points, uint64_t n, uint32_t dims, int device, pcpx_kd_index** out);
void pcpx_kd_destroy(pcpx_kd_index* idx);
uint64_t pcpx_kd_size(const pcpx_kd_index* idx);
uint32_t pcpx_kd_dims(const pcpx_kd_index* idx);
int pcpx_kd_knn_batch(pcpx_kd_index* idx, const float* queries, uint64_t nq, uint32_t k, float eps, uint32_t* out_idx, uint32_t* out_count,
                      float* opt_out_d2);
int pcpx_kd_range_aabb_batch(pcpx_kd_index* idx, const float* boxes, uint64_t nb, uint64_t* out_offsets, uint32_t* out_idx,
                             uint64_t idx_capacity);

#ifdef __cplusplus
}
#endif
#endif /* PCPX_H */
