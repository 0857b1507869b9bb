/*
 * ndt_synth.h - device-side generator of the synthetic NDT workloads (libndt_synth.so).
 *
 * NOT part of the matcher and not part of the drop-in boundary: this is the workload generator
 * SURVEY.md section 8d asks for ("RNG identical in Python, C++ and HIP so large inputs are generated
 * in place from a seed, never shipped"; config 4: "Generated on device").  It reproduces
 * gtsam_ndt_amd/synth.py bit for bit - splitmix64 counters, float64 arithmetic restricted to the
 * exactly-rounded operations (+ - * / sqrt), no contraction - so that bench.py can fill the 4096
 * loop-closure candidates of BASELINE config 4 (6.55 GB) in HBM in tens of milliseconds instead
 * of minutes of numpy, and the tests can check any pair of them against the numpy generator.
 * The reference checkout holds no data or generator (/root/reference/README.md:1).
 */
#ifndef NDT_SYNTH_H_
#define NDT_SYNTH_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NDT_SYNTH_ROOM_SEGMENTS 76 /* 4 walls + 24 segments + 12 boxes/diamonds x 4 (synth.room_scene) */

/* synth.room_scene(seed, L, x0, y0): segments[k] = (ax, ay, bx, by), k < 76.  Host only, no device. */
int32_t ndt_synth_room_scene(uint64_t seed, double L, double x0, double y0, double* segments /*[76][4]*/);

/* synth.sample_scene(scene, n, seed, sigma, first) followed - when pose != NULL - by
 * synth.to_source_frame(x, y, pose), rounded to float32: n points on the device.  cs_sn = (cos, sin)
 * of pose[2] as the caller's libm computes them (the one inexact operation of the generator is
 * left to the host so that every implementation shares it).  segments is a HOST array [n_seg][4];
 * d_x / d_y are device arrays of n floats.  Asynchronous on `stream` once the scene is uploaded. */
int32_t ndt_synth_sample_dev(const double* segments, int32_t n_seg, size_t n, uint64_t seed, double sigma, uint64_t first,
                             const double* pose /*[3] or NULL*/, const double* cs_sn /*[2] or NULL*/, float* d_x,
                             float* d_y, void* stream);

/* The candidates first_pair .. first_pair + n_pairs - 1 of BASELINE config 4, exactly as
 * synth.make_pair(4, pair_index = k, n_tgt, n_src, sigma) builds them, laid out as
 * ndt2d_batch_align_dev takes them: d_tx/d_ty [n_pairs * n_tgt], d_sx/d_sy [n_pairs * n_src],
 * d_toff/d_soff [n_pairs + 1] (uint64), d_init [n_pairs][3] (the initial guesses, all zero for this
 * config), d_pose [n_pairs][3] (the generating poses; may be NULL).  All pointers are device
 * pointers.  Returns when everything is enqueued on `stream`. */
int32_t ndt_synth_config4_dev(uint64_t first_pair, size_t n_pairs, size_t n_tgt, size_t n_src, double sigma, float* d_tx,
                              float* d_ty, float* d_sx, float* d_sy, uint64_t* d_toff, uint64_t* d_soff, double* d_init,
                              double* d_pose, void* stream);

/* synth3d.lidar_scan(seed, pose, n_elev, n_azim, sigma) on the device: the 64-beam lidar of BASELINE config 5 ray
 * cast against the box room [-L/2, L/2]^2 x [0, height] with the clutter boxes boxes_lo / boxes_hi (HOST arrays
 * [n_box][3], n_box <= 32: synth3d.scene_boxes), sensor at pose (tx, ty, tz + sensor_z; roll, pitch, yaw), points in
 * the SENSOR frame, ring by ring (firing_order = 0: index = ring * n_azim + bearing, as synth3d) or in firing order
 * (firing_order = 1: index = bearing * n_elev + ring - all beams of one bearing, then the next bearing, as a
 * spinning lidar's driver delivers them; the same points, permuted).  Same beams, same noise counters, float64 ray casting; NOT bit-identical to numpy
 * (the beams' cos / sin come from the device's libm): points agree to float32 rounding except for a ray that grazes
 * a box edge.  d_x / d_y / d_z: device arrays of n_elev * n_azim floats.  Asynchronous on `stream`. */
#define NDT_SYNTH_MAX_BOXES 32
int32_t ndt_synth_lidar3d_dev(const double* boxes_lo, const double* boxes_hi, int32_t n_box, double L, double height,
                              double sensor_z, uint64_t seed, const double pose[6], int32_t n_elev, int32_t n_azim,
                              double sigma, int32_t firing_order, float* d_x, float* d_y, float* d_z, void* stream);

/* text of the last error on this thread ("" if none) */
const char* ndt_synth_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* NDT_SYNTH_H_ */
