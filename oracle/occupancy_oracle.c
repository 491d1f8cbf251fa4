/* TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * Plain-C restatement of karto::OccupancyGrid::CreateFromScans:
 *   CreateFromScans   Karto.h:6118-6139      AddScan    Karto.h:6148-6189
 *   RayTrace          Karto.h:6199-6232      TraceLine  Karto.h:4874-4927 (Bresenham over Grid<kt_int32u>)
 *   UpdateCell/Update Karto.h:6240-6274      WorldToGrid Karto.h:4421-4436, Math.h:87-90
 *   Grid::Resize width step = align8(width), Karto.h:4640
 * PINNED against the reference build: tests/golden/occupancy.npz holds the cells and both counter grids the
 * reference's own OccupancyGrid produced (tests/golden/make_golden_occupancy.py). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {int32_t n; const double * ranges; const double * points; double sensor_pose[3];} ko_scan;

static double occ_round(double v) {return v >= 0.0 ? floor(v + 0.5) : ceil(v - 0.5);}   /* Math.h:87-90 */
static int32_t occ_to_int(double v) {if (!(v > -2147483649.0 && v < 2147483648.0)) {return INT32_MIN;} return (int32_t)v;}

static void occ_trace(uint32_t * pass, int32_t w, int32_t h, int32_t ws, int32_t x0, int32_t y0, int32_t x1, int32_t y1)
{
  /* Grid<T>::TraceLine, Karto.h:4874-4927 */
  const int steep = abs(y1 - y0) > abs(x1 - x0);
  int32_t t;
  if (steep) {t = x0; x0 = y0; y0 = t; t = x1; x1 = y1; y1 = t;}
  if (x0 > x1) {t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t;}
  const int32_t deltaX = x1 - x0, deltaY = abs(y1 - y0);
  int32_t error = 0, y = y0;
  const int32_t ystep = y0 < y1 ? 1 : -1;
  for (int32_t x = x0; x <= x1; x++) {
    const int32_t px = steep ? y : x, py = steep ? x : y;
    error += deltaY;
    if (2 * error >= deltaX) {y += ystep; error -= deltaX;}
    if (px >= 0 && px < w && py >= 0 && py < h) {pass[px + py * ws]++;}
  }
}

/* pass, hits: ws*height uint32 (zeroed here); cells: ws*height uint8.  ws = align8(width). */
void ko_occupancy_from_scans(
  int32_t width, int32_t height, double off_x, double off_y, double resolution, int32_t n_scans, const ko_scan * scans,
  double range_threshold, double min_range, double max_range, uint32_t min_pass_through, double occupancy_threshold,
  uint32_t * pass, uint32_t * hits, uint8_t * cells)
{
  const int32_t ws = (width + 7) & ~7;
  const size_t size = (size_t)ws * height;
  const double scale = 1.0 / resolution;
  memset(pass, 0, size * 4); memset(hits, 0, size * 4);
  for (int32_t s = 0; s < n_scans; ++s) {
    const ko_scan * sc = &scans[s];
    const double sx = sc->sensor_pose[0], sy = sc->sensor_pose[1];
    for (int32_t i = 0; i < sc->n; ++i) {
      double px = sc->points[2 * i], py = sc->points[2 * i + 1];
      const double r = sc->ranges[i];
      const int valid_end = r < (range_threshold - 1e-06);
      if (r <= min_range || r >= max_range || isnan(r)) {continue;}
      if (r >= range_threshold) {
        const double ratio = range_threshold / r;
        const double dx = px - sx, dy = py - sy;
        px = sx + ratio * dx; py = sy + ratio * dy;
      }
      const int32_t fx = occ_to_int(occ_round((sx - off_x) * scale)), fy = occ_to_int(occ_round((sy - off_y) * scale));
      const int32_t tx = occ_to_int(occ_round((px - off_x) * scale)), ty = occ_to_int(occ_round((py - off_y) * scale));
      occ_trace(pass, width, height, ws, fx, fy, tx, ty);
      if (valid_end && tx >= 0 && tx < width && ty >= 0 && ty < height) {pass[tx + ty * ws]++; hits[tx + ty * ws]++;}
    }
  }
  memset(cells, 0, size);                                       /* GridStates_Unknown */
  for (size_t k = 0; k < size; ++k) {
    if (pass[k] > min_pass_through) {
      const double ratio = (double)hits[k] / (double)pass[k];
      cells[k] = ratio > occupancy_threshold ? 100 : 255;       /* GridStates_Occupied / GridStates_Free */
    }
  }
}
