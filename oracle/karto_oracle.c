/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of slam_toolbox's karto correlative scan matcher (hot path A of
 * SURVEY.md section 8).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this; the product library (slam_toolbox_amd/csrc) never does.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit against the reference's
 * own sources compiled in place (oracle/_ref, oracle/ref_driver.cpp) by
 * tests/test_oracle_vs_reference.py (dev container) and against the committed fixtures made
 * from that build (tests/golden/, tests/test_oracle_golden.py; those run everywhere).
 *
 * Each function cites the reference file:line it restates (paths relative to /root/reference;
 * Mapper.cpp = lib/karto_sdk/src/Mapper.cpp, Karto.h/Mapper.h/Math.h under
 * lib/karto_sdk/include/karto_sdk/).  Build with -ffp-contract=off: the reference ships as
 * generic x86-64 (no FMA contraction) and the grid-index roundings below are contraction
 * sensitive.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define KO_TOLERANCE 1e-06            /* Math.h:41 KT_TOLERANCE */
#define KO_INVALID_SCAN INT32_MAX     /* Math.h:47 */
#define KO_PI 3.14159265358979323846  /* Math.h KT_PI */
#define KO_2PI 6.28318530717958647692
#define KO_PI_180 0.01745329251994329577 /* Math.h:35 KT_PI_180 */
#define KO_OCCUPIED 100               /* Karto.h GridStates_Occupied */
#define KO_MAX_VARIANCE 500.0         /* Mapper.cpp:52 */
#define KO_DISTANCE_PENALTY_GAIN 0.2  /* Mapper.cpp:53 */
#define KO_ANGLE_PENALTY_GAIN 0.2     /* Mapper.cpp:54 */

typedef struct
{
  int32_t n;              /* number of range readings (beams) */
  const double * ranges;  /* n */
  const double * points;  /* 2n: UNFILTERED world points (Karto.h:5613-5628, wantFiltered=false) */
  double sensor_pose[3];
} ko_scan;

typedef struct
{
  double response;
  double x, y, heading;
} ko_pose_response;      /* std::pair<kt_double, Pose2>, Mapper.cpp:761 */

typedef struct
{
  /* CorrelationGrid (Mapper.h:1074-1314) over Grid<kt_int8u> (Karto.h:4573-4965) */
  int32_t width, height, width_step, data_size;
  int32_t roi_x, roi_y, roi_w, roi_h;
  int32_t kernel_size;
  uint8_t * data;
  uint8_t * kernel;
  double scale, off_x, off_y;  /* CoordinateConverter (Karto.h:4393-4563) */
  double smear;
  /* search space probs: Grid<kt_double> side x side (Mapper.cpp:513-514) */
  int32_t probs_side, probs_ws;
  double * probs;
  double probs_scale, probs_off_x, probs_off_y;
  /* GridIndexLookup (Karto.h:6603-6963) */
  int32_t n_angles, n_points, lookup_cap;
  int32_t * lookup;   /* n_angles x n_points */
  double * angles;
  /* the eight Mapper parameters ScanMatcher reads (Mapper.cpp:590-627, 671-682) */
  double coarse_search_angle_offset, coarse_angle_resolution, fine_search_angle_offset;
  int use_response_expansion;
  double distance_variance_penalty, minimum_distance_penalty;  /* variance penalties AS STORED (squared) */
  double angle_variance_penalty, minimum_angle_penalty;
  /* last CorrelateScan response volume, kept for the tests (reference frees it, Mapper.cpp:832) */
  ko_pose_response * volume;
  int32_t vol_nx, vol_ny, vol_na;
  int n_threads;
} ko_matcher;

/* Math.h:87-90 */
static double ko_round(double v) {return v >= 0.0 ? floor(v + 0.5) : ceil(v - 0.5);}

/* static_cast<kt_int32s>(double) as x86-64 cvttsd2si does it: out of range / NaN -> INT32_MIN */
static int32_t ko_to_int(double v)
{
  if (!(v > -2147483649.0 && v < 2147483648.0)) {return INT32_MIN;}
  return (int32_t)v;
}

/* Math.h:135-139 */
static int ko_double_equal(double a, double b)
{
  double delta = a - b;
  return delta < 0.0 ? delta >= -KO_TOLERANCE : delta <= KO_TOLERANCE;
}

/* Math.h:181-202 */
static double ko_normalize_angle(double angle)
{
  while (angle < -KO_PI) {
    if (angle < -KO_2PI) {
      angle += (uint32_t)(angle / -KO_2PI) * KO_2PI;
    } else {
      angle += KO_2PI;
    }
  }
  while (angle > KO_PI) {
    if (angle > KO_2PI) {
      angle -= (uint32_t)(angle / KO_2PI) * KO_2PI;
    } else {
      angle -= KO_2PI;
    }
  }
  return angle;
}

/* Math.h:213-224 */
static double ko_normalize_angle_difference(double minuend, double subtrahend)
{
  while (minuend - subtrahend < -KO_PI) {minuend += KO_2PI;}
  while (minuend - subtrahend > KO_PI) {minuend -= KO_2PI;}
  return minuend;
}

/* CoordinateConverter::WorldToGrid, Karto.h:4421-4436 (flipY = false) */
static void ko_world_to_grid(double scale, double ox, double oy, double wx, double wy, int32_t * gx, int32_t * gy)
{
  double gridX = (wx - ox) * scale;
  double gridY = (wy - oy) * scale;
  *gx = ko_to_int(ko_round(gridX));
  *gy = ko_to_int(ko_round(gridY));
}

static double ko_resolution(const ko_matcher * m) {return 1.0 / m->scale;} /* Karto.h:4518-4521 */

/* Mapper.h:1275-1280 */
static int32_t ko_half_kernel_size(double smear, double resolution)
{
  return (int32_t)ko_round(2.0 * smear / resolution);
}

/* LocalizedRangeScan::Update, Karto.h:5644-5704: unfiltered point of every beam */
void ko_scan_points(const double * ranges, int32_t n, const double * sensor_pose, double min_angle, double ang_res, double * out_xy)
{
  for (int32_t i = 0; i < n; i++) {
    double angle = sensor_pose[2] + min_angle + i * ang_res;
    out_xy[2 * i] = sensor_pose[0] + (ranges[i] * cos(angle));
    out_xy[2 * i + 1] = sensor_pose[1] + (ranges[i] * sin(angle));
  }
}

void ko_matcher_destroy(ko_matcher * m)
{
  if (!m) {return;}
  free(m->data); free(m->kernel); free(m->probs); free(m->lookup); free(m->angles); free(m->volume);
  free(m);
}

/* ScanMatcher::Create (Mapper.cpp:477-522) + CorrelationGrid ctor / CalculateKernel (Mapper.h:1099-1114,
 * 1194-1266) + Grid::Resize (Karto.h:4636-4664).  Returns NULL where the reference returns NULL or throws. */
ko_matcher * ko_matcher_create(double search_size, double resolution, double smear, double range_threshold)
{
  if (resolution <= 0) {return NULL;}
  if (search_size <= 0) {return NULL;}
  if (smear < 0) {return NULL;}
  if (range_threshold <= 0) {return NULL;}

  uint32_t side = (uint32_t)(ko_round(search_size / resolution) + 1);
  uint32_t margin = (uint32_t)ceil(range_threshold / resolution);
  int32_t grid_size = (int32_t)(side + 2 * margin);

  ko_matcher * m = (ko_matcher *)calloc(1, sizeof(ko_matcher));
  uint32_t border = (uint32_t)ko_half_kernel_size(smear, resolution) + 1;
  m->width = grid_size + 2 * (int32_t)border;
  m->height = grid_size + 2 * (int32_t)border;
  m->width_step = (int32_t)(((size_t)m->width + 7) & ~(size_t)7);   /* Math.h:233-237 */
  m->data_size = m->width_step * m->height;
  m->data = (uint8_t *)calloc((size_t)m->data_size, 1);
  m->scale = 1.0 / resolution;
  m->roi_x = (int32_t)border; m->roi_y = (int32_t)border; m->roi_w = grid_size; m->roi_h = grid_size;
  m->smear = smear;

  /* CalculateKernel, Mapper.h:1213-1266 */
  double res = ko_resolution(m);
  double min_dev = 0.5 * res, max_dev = 10 * res;
  if (!(smear >= min_dev && smear <= max_dev)) {ko_matcher_destroy(m); return NULL;}
  m->kernel_size = 2 * ko_half_kernel_size(smear, res) + 1;
  m->kernel = (uint8_t *)malloc((size_t)m->kernel_size * m->kernel_size);
  int32_t hk = m->kernel_size / 2;
  for (int32_t i = -hk; i <= hk; i++) {
    for (int32_t j = -hk; j <= hk; j++) {
      double d = hypot(i * res, j * res);
      double z = exp(-0.5 * pow(d / smear, 2));
      uint32_t v = (uint32_t)ko_round(z * KO_OCCUPIED);
      m->kernel[(i + hk) + m->kernel_size * (j + hk)] = (uint8_t)v;
    }
  }

  m->probs_side = (int32_t)side;
  m->probs_ws = (int32_t)(((size_t)side + 7) & ~(size_t)7);
  m->probs = (double *)calloc((size_t)m->probs_ws * side, sizeof(double));
  m->probs_scale = 1.0 / resolution;

  /* karto defaults, Mapper.cpp:2250-2293 */
  m->coarse_search_angle_offset = 20 * KO_PI_180;
  m->coarse_angle_resolution = 2 * KO_PI_180;
  m->fine_search_angle_offset = 0.2 * KO_PI_180;
  m->use_response_expansion = 0;
  m->distance_variance_penalty = 0.3 * 0.3;
  m->minimum_distance_penalty = 0.5;
  m->angle_variance_penalty = (20 * KO_PI_180) * (20 * KO_PI_180);
  m->minimum_angle_penalty = 0.9;
  m->n_threads = 1;
  return m;
}

/* values AS STORED in the Mapper parameters (i.e. after the setters squared the two variances,
 * Mapper.cpp:2562-2570) */
void ko_matcher_set_params(
  ko_matcher * m, double coarse_search_angle_offset, double coarse_angle_resolution,
  double fine_search_angle_offset, int use_response_expansion, double distance_variance_penalty,
  double minimum_distance_penalty, double angle_variance_penalty, double minimum_angle_penalty)
{
  m->coarse_search_angle_offset = coarse_search_angle_offset;
  m->coarse_angle_resolution = coarse_angle_resolution;
  m->fine_search_angle_offset = fine_search_angle_offset;
  m->use_response_expansion = use_response_expansion;
  m->distance_variance_penalty = distance_variance_penalty;
  m->minimum_distance_penalty = minimum_distance_penalty;
  m->angle_variance_penalty = angle_variance_penalty;
  m->minimum_angle_penalty = minimum_angle_penalty;
}

void ko_matcher_set_threads(ko_matcher * m, int n) {m->n_threads = n < 1 ? 1 : n;}

/* ScanMatcher::FindValidPoints, Mapper.cpp:1113-1164.  Returns the number of points written. */
int32_t ko_find_valid_points(const ko_scan * scan, const double * viewpoint, double * out_xy)
{
  const double min_sq = 0.1 * 0.1;
  int32_t trailing = 0, n_out = 0;
  double first_x = 0.0, first_y = 0.0;   /* Vector2 default ctor */
  int first_time = 1;
  for (int32_t it = 0; it < scan->n; it++) {
    double cx = scan->points[2 * it], cy = scan->points[2 * it + 1];
    if (first_time && !isnan(cx) && !isnan(cy)) {
      first_x = cx; first_y = cy; first_time = 0;
    }
    double dx = first_x - cx, dy = first_y - cy;
    if (dx * dx + dy * dy > min_sq) {
      double a = viewpoint[1] - first_y;
      double b = first_x - viewpoint[0];
      double c = first_y * viewpoint[0] - first_x * viewpoint[1];
      double ss = cx * a + cy * b + c;
      first_x = cx; first_y = cy;
      if (ss < 0.0) {
        trailing = it;
      } else {
        for (; trailing != it; ++trailing) {
          out_xy[2 * n_out] = scan->points[2 * trailing];
          out_xy[2 * n_out + 1] = scan->points[2 * trailing + 1];
          n_out++;
        }
      }
    }
  }
  return n_out;
}

/* CorrelationGrid::GridIndex (Mapper.h:1122-1128) over Grid::GridIndex (Karto.h:4681-4699), no bounds check */
static int32_t ko_grid_index(const ko_matcher * m, int32_t gx, int32_t gy)
{
  int32_t x = gx + m->roi_x, y = gy + m->roi_y;
  return x + (y * m->width_step);
}

/* CorrelationGrid::SmearPoint, Mapper.h:1152-1183 */
static void ko_smear_point(ko_matcher * m, int32_t gx, int32_t gy)
{
  int32_t gi = ko_grid_index(m, gx, gy);
  if (m->data[gi] != KO_OCCUPIED) {return;}
  int32_t hk = m->kernel_size / 2;
  for (int32_t j = -hk; j <= hk; j++) {
    uint8_t * row = m->data + ko_grid_index(m, gx, gy + j);
    int32_t kc = hk + m->kernel_size * (j + hk);
    for (int32_t i = -hk; i <= hk; i++) {
      uint8_t kv = m->kernel[i + kc];
      if (kv > row[i]) {row[i] = kv;}
    }
  }
}

/* ScanMatcher::AddScan, Mapper.cpp:1073-1105 */
static void ko_add_scan(ko_matcher * m, const ko_scan * scan, const double * viewpoint, double * tmp)
{
  int32_t nv = ko_find_valid_points(scan, viewpoint, tmp);
  for (int32_t k = 0; k < nv; k++) {
    int32_t gx, gy;
    ko_world_to_grid(m->scale, m->off_x, m->off_y, tmp[2 * k], tmp[2 * k + 1], &gx, &gy);
    if (!(gx >= 0 && gx < m->roi_w) || !(gy >= 0 && gy < m->roi_h)) {continue;}
    int32_t gi = ko_grid_index(m, gx, gy);
    if (m->data[gi] == KO_OCCUPIED) {continue;}
    m->data[gi] = KO_OCCUPIED;
    ko_smear_point(m, gx, gy);
  }
}

/* ScanMatcher::AddScans, Mapper.cpp:1032-1045 (Clear = Karto.h:4612-4615) */
void ko_add_scans(ko_matcher * m, const ko_scan * base, int32_t n_base, const double * viewpoint)
{
  memset(m->data, 0, (size_t)m->data_size);
  int32_t max_n = 0;
  for (int32_t s = 0; s < n_base; s++) {if (base[s].n > max_n) {max_n = base[s].n;}}
  double * tmp = (double *)malloc(sizeof(double) * 2 * (size_t)(max_n > 0 ? max_n : 1));
  for (int32_t s = 0; s < n_base; s++) {ko_add_scan(m, &base[s], viewpoint, tmp);}
  free(tmp);
}

/* MatchScan steps 1-4 (Mapper.cpp:543-569): centre the grid on the scan's sensor pose */
void ko_center_grid(ko_matcher * m, const double * scan_pose)
{
  double res = ko_resolution(m);
  m->off_x = scan_pose[0] - (0.5 * (m->roi_w - 1) * res);
  m->off_y = scan_pose[1] - (0.5 * (m->roi_h - 1) * res);
}

/* GridIndexLookup::ComputeOffsets, Karto.h:6797-6894 (+ Transform, Karto.h:2946-3041;
 * Matrix3::FromAxisAngle :2482-2511; Matrix3*Pose2 :2654-2666) */
void ko_compute_offsets(ko_matcher * m, const ko_scan * scan, double angle_center, double angle_offset, double angle_resolution)
{
  uint32_t n_angles = (uint32_t)(ko_round(angle_offset * 2.0 / angle_resolution) + 1);
  int32_t np = scan->n;
  if ((int64_t)n_angles * np > m->lookup_cap) {
    free(m->lookup); free(m->angles);
    m->lookup_cap = (int32_t)(n_angles * (uint32_t)np);
    m->lookup = (int32_t *)malloc(sizeof(int32_t) * (size_t)m->lookup_cap);
    m->angles = (double *)malloc(sizeof(double) * n_angles);
  } else {
    m->angles = (double *)realloc(m->angles, sizeof(double) * (n_angles ? n_angles : 1));
  }
  m->n_angles = (int32_t)n_angles;
  m->n_points = np;

  /* Transform(sensorPose): rotation about z by (0 - heading); m_Transform = pose */
  double th = scan->sensor_pose[2];
  double tx = scan->sensor_pose[0], ty = scan->sensor_pose[1];
  double r00, r01, r02, r10, r11, r12;
  if (tx == 0.0 && ty == 0.0 && th == 0.0) {
    r00 = 1; r01 = 0; r02 = 0; r10 = 0; r11 = 1; r12 = 0;   /* Pose2() == pose: identity, Karto.h:3006-3011 */
  } else {
    double radians = 0.0 - th;
    double cosR = cos(radians), sinR = sin(radians), omc = 1.0 - cosR;
    r00 = 0.0 * omc + cosR;
    r01 = 0.0 * 0.0 * omc - 1.0 * sinR;
    r02 = 0.0 * 1.0 * omc + 0.0 * sinR;
    r10 = 0.0 * 0.0 * omc + 1.0 * sinR;
    r11 = 0.0 * omc + cosR;
    r12 = 0.0 * 1.0 * omc - 0.0 * sinR;
  }
  double * local = (double *)malloc(sizeof(double) * 2 * (size_t)(np > 0 ? np : 1));
  for (int32_t i = 0; i < np; i++) {
    double sx = scan->points[2 * i] - tx, sy = scan->points[2 * i + 1] - ty, sh = 0.0 - th;
    local[2 * i] = r00 * sx + r01 * sy + r02 * sh;
    local[2 * i + 1] = r10 * sx + r11 * sy + r12 * sh;
  }

  double start_angle = angle_center - angle_offset;
  for (uint32_t a = 0; a < n_angles; a++) {
    double angle = start_angle + a * angle_resolution;
    m->angles[a] = angle;
    double cosine = cos(angle), sine = sin(angle);
    int32_t * row = m->lookup + (size_t)a * np;
    for (int32_t i = 0; i < np; i++) {
      if (isnan(scan->ranges[i]) || isinf(scan->ranges[i])) {
        row[i] = KO_INVALID_SCAN;
        continue;
      }
      double ox = cosine * local[2 * i] - sine * local[2 * i + 1];
      double oy = sine * local[2 * i] + cosine * local[2 * i + 1];
      int32_t gx, gy;
      ko_world_to_grid(m->scale, m->off_x, m->off_y, ox + m->off_x, oy + m->off_y, &gx, &gy);
      /* base Grid::GridIndex, no ROI, no bounds check (Karto.h:6886-6887); int32 wrap-around as compiled */
      row[i] = (int32_t)((uint32_t)gx + (uint32_t)gy * (uint32_t)m->width_step);
    }
  }
  free(local);
}

/* ScanMatcher::GetResponse, Mapper.cpp:1172-1208 */
double ko_get_response(const ko_matcher * m, uint32_t angle_index, int32_t grid_position_index)
{
  double response = 0.0;
  const uint8_t * byte = m->data + grid_position_index;
  const int32_t * off = m->lookup + (size_t)angle_index * m->n_points;
  uint32_t n_points = (uint32_t)m->n_points;
  if (n_points == 0) {return response;}
  for (uint32_t i = 0; i < n_points; i++) {
    int32_t pgi = (int32_t)((uint32_t)grid_position_index + (uint32_t)off[i]);
    if (!(pgi >= 0 && pgi < m->data_size) || off[i] == KO_INVALID_SCAN) {continue;}
    response += byte[off[i]];
  }
  response /= (n_points * KO_OCCUPIED);
  return response;
}

typedef struct
{
  const ko_matcher * m;
  const double * x_poses; const double * y_poses;
  uint32_t nx, ny, na;
  double cx, cy, ch, angle_offset, angle_resolution;
  int do_penalize;
  ko_pose_response * out;
  uint32_t y_begin, y_step;
} ko_row_job;

/* ScanMatcher::operator()(y), Mapper.cpp:641-694 */
static void ko_score_row(const ko_row_job * j, uint32_t y_pose)
{
  const ko_matcher * m = j->m;
  double y = j->y_poses[y_pose];
  double newPositionY = j->cy + y;
  double squareY = y * y;
  for (uint32_t x_pose = 0; x_pose < j->nx; x_pose++) {
    double x = j->x_poses[x_pose];
    double newPositionX = j->cx + x;
    double squareX = x * x;
    int32_t gx, gy;
    ko_world_to_grid(m->scale, m->off_x, m->off_y, newPositionX, newPositionY, &gx, &gy);
    int32_t gridIndex = ko_grid_index(m, gx, gy);
    double startAngle = j->ch - j->angle_offset;
    for (uint32_t ai = 0; ai < j->na; ai++) {
      double angle = startAngle + ai * j->angle_resolution;
      double response = ko_get_response(m, ai, gridIndex);
      if (j->do_penalize && (ko_double_equal(response, 0.0) == 0)) {
        double squaredDistance = squareX + squareY;
        double distancePenalty = 1.0 - (KO_DISTANCE_PENALTY_GAIN * squaredDistance / m->distance_variance_penalty);
        distancePenalty = distancePenalty > m->minimum_distance_penalty ? distancePenalty : m->minimum_distance_penalty;
        double sad = (angle - j->ch) * (angle - j->ch);
        double anglePenalty = 1.0 - (KO_ANGLE_PENALTY_GAIN * sad / m->angle_variance_penalty);
        anglePenalty = anglePenalty > m->minimum_angle_penalty ? anglePenalty : m->minimum_angle_penalty;
        response *= (distancePenalty * anglePenalty);
      }
      ko_pose_response * o = &j->out[(y_pose * j->nx + x_pose) * j->na + ai];
      o->response = response; o->x = newPositionX; o->y = newPositionY; o->heading = ko_normalize_angle(angle);
    }
  }
}

static void * ko_row_worker(void * arg)
{
  const ko_row_job * j = (const ko_row_job *)arg;
  for (uint32_t y = j->y_begin; y < j->ny; y += j->y_step) {ko_score_row(j, y);}
  return NULL;
}

/* ScanMatcher::ComputePositionalCovariance, Mapper.cpp:874-966 */
static void ko_positional_covariance(
  ko_matcher * m, const double * best_pose, double best_response, const double * center,
  double off_x, double off_y, double res_x, double res_y, double angle_res, double * cov)
{
  memset(cov, 0, sizeof(double) * 9);
  cov[0] = 1.0; cov[4] = 1.0; cov[8] = 1.0;
  if (best_response < KO_TOLERANCE) {
    cov[0] = KO_MAX_VARIANCE; cov[4] = KO_MAX_VARIANCE; cov[8] = 4 * (angle_res * angle_res);
    return;
  }
  double aXX = 0, aXY = 0, aYY = 0, norm = 0;
  double dx = best_pose[0] - center[0], dy = best_pose[1] - center[1];
  uint32_t nX = (uint32_t)(ko_round(off_x * 2.0 / res_x) + 1);
  double startX = -off_x;
  uint32_t nY = (uint32_t)(ko_round(off_y * 2.0 / res_y) + 1);
  double startY = -off_y;
  for (uint32_t yi = 0; yi < nY; yi++) {
    double y = startY + yi * res_y;
    for (uint32_t xi = 0; xi < nX; xi++) {
      double x = startX + xi * res_x;
      int32_t gx, gy;
      ko_world_to_grid(m->probs_scale, m->probs_off_x, m->probs_off_y, center[0] + x, center[1] + y, &gx, &gy);
      double response = m->probs[gx + gy * m->probs_ws];
      if (response >= (best_response - 0.1)) {
        norm += response;
        aXX += ((x - dx) * (x - dx) * response);
        aXY += ((x - dx) * (y - dy) * response);
        aYY += ((y - dy) * (y - dy) * response);
      }
    }
  }
  if (norm > KO_TOLERANCE) {
    double vXX = aXX / norm, vXY = aXY / norm, vYY = aYY / norm;
    double vTHTH = 4 * (angle_res * angle_res);
    double minXX = 0.1 * (res_x * res_x), minYY = 0.1 * (res_y * res_y);
    vXX = vXX > minXX ? vXX : minXX;
    vYY = vYY > minYY ? vYY : minYY;
    double mult = 1.0 / best_response;
    cov[0] = vXX * mult; cov[1] = vXY * mult; cov[3] = vXY * mult; cov[4] = vYY * mult; cov[8] = vTHTH;
  }
  if (ko_double_equal(cov[0], 0.0)) {cov[0] = KO_MAX_VARIANCE;}
  if (ko_double_equal(cov[4], 0.0)) {cov[4] = KO_MAX_VARIANCE;}
}

/* ScanMatcher::ComputeAngularCovariance, Mapper.cpp:977-1025 */
static void ko_angular_covariance(
  ko_matcher * m, const double * best_pose, double best_response, const double * center,
  double angle_offset, double angle_res, double * cov)
{
  double bestAngle = ko_normalize_angle_difference(best_pose[2], center[2]);
  int32_t gx, gy;
  ko_world_to_grid(m->scale, m->off_x, m->off_y, best_pose[0], best_pose[1], &gx, &gy);
  int32_t gridIndex = ko_grid_index(m, gx, gy);
  uint32_t nAngles = (uint32_t)(ko_round(angle_offset * 2 / angle_res) + 1);
  double startAngle = center[2] - angle_offset;
  double norm = 0.0, acc = 0.0;
  for (uint32_t ai = 0; ai < nAngles; ai++) {
    double angle = startAngle + ai * angle_res;
    double response = ko_get_response(m, ai, gridIndex);
    if (response >= (best_response - 0.1)) {
      norm += response;
      acc += ((angle - bestAngle) * (angle - bestAngle) * response);
    }
  }
  if (norm > KO_TOLERANCE) {
    if (acc < KO_TOLERANCE) {acc = angle_res * angle_res;}
    acc /= norm;
  } else {
    acc = 1000 * (angle_res * angle_res);
  }
  cov[8] = acc;
}

/* ScanMatcher::CorrelateScan, Mapper.cpp:712-862.  cov is in/out (the fine pass keeps the coarse xy block).
 * Returns the response, or -1e9 where the reference throws. */
double ko_correlate_scan(
  ko_matcher * m, const ko_scan * scan, const double * center, double off_x, double off_y,
  double res_x, double res_y, double angle_offset, double angle_res, int do_penalize,
  double * mean, double * cov, int fine)
{
  ko_compute_offsets(m, scan, center[2], angle_offset, angle_res);

  if (!fine) {
    memset(m->probs, 0, sizeof(double) * (size_t)m->probs_ws * m->probs_side);
    m->probs_off_x = center[0] - off_x;
    m->probs_off_y = center[1] - off_y;
  }

  uint32_t nX = (uint32_t)(ko_round(off_x * 2.0 / res_x) + 1);
  double startX = -off_x;
  double * x_poses = (double *)malloc(sizeof(double) * (nX ? nX : 1));
  for (uint32_t i = 0; i < nX; i++) {x_poses[i] = startX + i * res_x;}
  uint32_t nY = (uint32_t)(ko_round(off_y * 2.0 / res_y) + 1);
  double startY = -off_y;
  double * y_poses = (double *)malloc(sizeof(double) * (nY ? nY : 1));
  for (uint32_t i = 0; i < nY; i++) {y_poses[i] = startY + i * res_y;}
  uint32_t nAngles = (uint32_t)(ko_round(angle_offset * 2.0 / angle_res) + 1);
  uint32_t size = nX * nY * nAngles;

  free(m->volume);
  m->volume = (ko_pose_response *)malloc(sizeof(ko_pose_response) * (size ? size : 1));
  m->vol_nx = (int32_t)nX; m->vol_ny = (int32_t)nY; m->vol_na = (int32_t)nAngles;

  /* tbb::parallel_for_each(m_yPoses, *this), Mapper.cpp:773 */
  int nt = m->n_threads;
  if (nt > (int)nY) {nt = (int)nY;}
  if (nt < 1) {nt = 1;}
  ko_row_job * jobs = (ko_row_job *)malloc(sizeof(ko_row_job) * (size_t)nt);
  pthread_t * tids = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nt);
  for (int t = 0; t < nt; t++) {
    ko_row_job j = {m, x_poses, y_poses, nX, nY, nAngles, center[0], center[1], center[2],
      angle_offset, angle_res, do_penalize, m->volume, (uint32_t)t, (uint32_t)nt};
    jobs[t] = j;
  }
  if (nt == 1) {
    ko_row_worker(&jobs[0]);
  } else {
    for (int t = 0; t < nt; t++) {pthread_create(&tids[t], NULL, ko_row_worker, &jobs[t]);}
    for (int t = 0; t < nt; t++) {pthread_join(tids[t], NULL);}
  }
  free(jobs); free(tids);

  double bestResponse = -1;
  int failed = 0;
  for (uint32_t i = 0; i < size; i++) {
    bestResponse = bestResponse > m->volume[i].response ? bestResponse : m->volume[i].response;
    if (!fine) {
      int32_t gx, gy;
      ko_world_to_grid(m->probs_scale, m->probs_off_x, m->probs_off_y, m->volume[i].x, m->volume[i].y, &gx, &gy);
      if (!(gx >= 0 && gx < m->probs_side) || !(gy >= 0 && gy < m->probs_side)) {failed = 1; break;}  /* :786-796 */
      double * ptr = &m->probs[gx + gy * m->probs_ws];
      *ptr = m->volume[i].response > *ptr ? m->volume[i].response : *ptr;
    }
  }
  free(x_poses); free(y_poses);
  if (failed) {return -1e9;}

  double ax = 0.0, ay = 0.0, thetaX = 0.0, thetaY = 0.0;
  int32_t count = 0;
  for (uint32_t i = 0; i < size; i++) {
    if (ko_double_equal(m->volume[i].response, bestResponse)) {
      ax += m->volume[i].x; ay += m->volume[i].y;
      double heading = m->volume[i].heading;
      thetaX += cos(heading);
      thetaY += sin(heading);
      count++;
    }
  }
  double avg[3];
  if (count > 0) {
    ax /= count; ay /= count;
    thetaX /= count; thetaY /= count;
    avg[0] = ax; avg[1] = ay; avg[2] = atan2(thetaY, thetaX);
  } else {
    return -1e9;   /* :828 */
  }

  if (!fine) {
    ko_positional_covariance(m, avg, bestResponse, center, off_x, off_y, res_x, res_y, angle_res, cov);
  } else {
    ko_angular_covariance(m, avg, bestResponse, center, angle_offset, angle_res, cov);
  }
  mean[0] = avg[0]; mean[1] = avg[1]; mean[2] = avg[2];
  if (bestResponse > 1.0) {bestResponse = 1.0;}
  return bestResponse;
}

/* ScanMatcher::MatchScan, Mapper.cpp:534-639 */
double ko_match_scan(
  ko_matcher * m, const ko_scan * scan, const ko_scan * base, int32_t n_base,
  int do_penalize, int do_refine, double * mean, double * cov)
{
  double scanPose[3] = {scan->sensor_pose[0], scan->sensor_pose[1], scan->sensor_pose[2]};
  memset(cov, 0, sizeof(double) * 9);  /* Matrix3 default ctor */
  if (scan->n == 0) {
    mean[0] = scanPose[0]; mean[1] = scanPose[1]; mean[2] = scanPose[2];
    cov[0] = KO_MAX_VARIANCE; cov[4] = KO_MAX_VARIANCE;
    cov[8] = 4 * (m->coarse_angle_resolution * m->coarse_angle_resolution);
    return 0.0;
  }
  ko_center_grid(m, scanPose);
  ko_add_scans(m, base, n_base, scanPose);

  double res = ko_resolution(m);
  double cso_x = 0.5 * ((double)m->probs_side - 1) * res, cso_y = 0.5 * ((double)m->probs_side - 1) * res;
  double csr = 2 * res;

  double best = ko_correlate_scan(m, scan, scanPose, cso_x, cso_y, csr, csr,
      m->coarse_search_angle_offset, m->coarse_angle_resolution, do_penalize, mean, cov, 0);
  if (best == -1e9) {return best;}

  if (m->use_response_expansion) {
    if (ko_double_equal(best, 0.0)) {
      double newOffset = m->coarse_search_angle_offset;
      for (uint32_t i = 0; i < 3; i++) {
        newOffset += 20 * KO_PI_180;   /* math::DegreesToRadians(20), Math.h:58 (KT_PI_180) */
        best = ko_correlate_scan(m, scan, scanPose, cso_x, cso_y, csr, csr, newOffset,
            m->coarse_angle_resolution, do_penalize, mean, cov, 0);
        if (best == -1e9) {return best;}
        if (ko_double_equal(best, 0.0) == 0) {break;}
      }
    }
  }

  if (do_refine) {
    double fso_x = csr * 0.5, fso_y = csr * 0.5;
    double center[3] = {mean[0], mean[1], mean[2]};
    best = ko_correlate_scan(m, scan, center, fso_x, fso_y, res, res,
        0.5 * m->coarse_angle_resolution, m->fine_search_angle_offset, do_penalize, mean, cov, 1);
  }
  return best;
}

/* ---- accessors for the tests ---- */
void ko_grid_info(const ko_matcher * m, int32_t * out, double * offset_scale)
{
  out[0] = m->width; out[1] = m->height; out[2] = m->width_step; out[3] = m->roi_x; out[4] = m->roi_y;
  out[5] = m->roi_w; out[6] = m->roi_h; out[7] = m->kernel_size; out[8] = m->data_size;
  offset_scale[0] = m->off_x; offset_scale[1] = m->off_y; offset_scale[2] = m->scale;
}
const uint8_t * ko_grid_data(const ko_matcher * m) {return m->data;}
const uint8_t * ko_kernel_data(const ko_matcher * m) {return m->kernel;}
const int32_t * ko_lookup_data(const ko_matcher * m, int32_t * n_angles, int32_t * n_points)
{
  *n_angles = m->n_angles; *n_points = m->n_points;
  return m->lookup;
}
const double * ko_probs_data(const ko_matcher * m, int32_t * side, int32_t * ws)
{
  *side = m->probs_side; *ws = m->probs_ws;
  return m->probs;
}
const ko_pose_response * ko_volume(const ko_matcher * m, int32_t * nx, int32_t * ny, int32_t * na)
{
  *nx = m->vol_nx; *ny = m->vol_ny; *na = m->vol_na;
  return m->volume;
}
int32_t ko_world_to_grid_index(const ko_matcher * m, double x, double y)
{
  int32_t gx, gy;
  ko_world_to_grid(m->scale, m->off_x, m->off_y, x, y, &gx, &gy);
  return ko_grid_index(m, gx, gy);
}
