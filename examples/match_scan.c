/* Minimal C caller of the C ABI (include/karto_hip.h): what a cgo / JNI / plain-C binding of the matcher looks
 * like.  Builds a tiny synthetic room, matches a displaced scan against two base scans with the karto default
 * preset and prints response, pose and covariance.
 *   gcc -std=c99 -O2 -I include examples/match_scan.c -L slam_toolbox_amd -lkartohip -lm \
 *       -Wl,-rpath,$PWD/slam_toolbox_amd -o /tmp/match_scan && /tmp/match_scan                      */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "karto_hip.h"

#define N_BEAMS 361

/* range of a beam from (x, y) at angle a inside the axis-aligned room [0, 8] x [0, 6] */
static double room_range(double x, double y, double a)
{
  const double c = cos(a), s = sin(a);
  double best = 1e9, t;
  if (c > 1e-12) {t = (8.0 - x) / c; if (t < best) {best = t;}}
  if (c < -1e-12) {t = (0.0 - x) / c; if (t < best) {best = t;}}
  if (s > 1e-12) {t = (6.0 - y) / s; if (t < best) {best = t;}}
  if (s < -1e-12) {t = (0.0 - y) / s; if (t < best) {best = t;}}
  return best;
}

static void make_scan(kh_scan * out, double * ranges, double * points, double x, double y, double th, double px,
  double py, double pth)
{
  /* readings taken at the TRUE pose (x, y, th); the scan object carries the (possibly wrong) pose (px, py, pth) */
  const double min_angle = -3.14159265358979323846 / 2.0, ang_res = 3.14159265358979323846 / (N_BEAMS - 1);
  for (int i = 0; i < N_BEAMS; ++i) {ranges[i] = room_range(x, y, th + min_angle + i * ang_res);}
  out->n = N_BEAMS; out->ranges = ranges; out->points_xy = points;
  out->device_points_xy = NULL;                  /* not kept resident on the device: uploaded with the call */
  out->sensor_pose[0] = px; out->sensor_pose[1] = py; out->sensor_pose[2] = pth;
  kh_scan_points(ranges, N_BEAMS, out->sensor_pose, min_angle, ang_res, points);   /* LocalizedRangeScan::Update */
}

int main(void)
{
  if (kh_device_count() < 1) {fprintf(stderr, "no GPU: %s\n", kh_version()); return 2;}
  kh_matcher * m = NULL;
  int rc = kh_matcher_create(0.3, 0.01, 0.03, 12.0, 0, 1, &m);        /* karto defaults, Mapper.cpp:2209-2225 */
  if (rc != KH_OK) {fprintf(stderr, "create failed: %d %s\n", rc, kh_last_error()); return 1;}
  kh_match_params p;
  kh_match_params_default(&p);
  kh_matcher_set_params(m, &p);
  static double r[3][N_BEAMS], pts[3][2 * N_BEAMS];
  kh_scan base[2], query;
  make_scan(&base[0], r[0], pts[0], 3.0, 2.0, 0.3, 3.0, 2.0, 0.3);
  make_scan(&base[1], r[1], pts[1], 3.2, 2.1, 0.35, 3.2, 2.1, 0.35);
  make_scan(&query, r[2], pts[2], 3.4, 2.2, 0.4, 3.45, 2.17, 0.42);      /* pose off by (5 cm, -3 cm, 0.02 rad) */
  double mean[3], cov[9], response = 0.0;
  rc = kh_matcher_match(m, &query, base, 2, 1, 1, mean, cov, &response);
  if (rc != KH_OK) {fprintf(stderr, "match failed: %d %s\n", rc, kh_last_error()); return 1;}
  printf("response %.6f pose %.4f %.4f %.5f cov_xx %.3e cov_yy %.3e cov_tt %.3e\n", response, mean[0], mean[1], mean[2],
    cov[0], cov[4], cov[8]);
  kh_matcher_destroy(m);
  /* the matched pose must be closer to the truth (3.4, 2.2, 0.4) than the prior */
  const double before = hypot(3.45 - 3.4, 2.17 - 2.2), after = hypot(mean[0] - 3.4, mean[1] - 2.2);
  return (response > 0.5 && after < before) ? 0 : 3;
}
