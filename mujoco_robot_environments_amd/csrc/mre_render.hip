// mre_render.hip -- batched overhead camera: depth, RGB and segmentation
// of every environment by casting one ray per pixel against the scene's boxes and ground plane.
// Replaces the three mujoco.Renderer passes of the reference (tasks/rearrangement.py:254-280 bounding
// boxes from the segmentation image, :460-478 rgb + depth observation, :500-530 depth look-up of
// pixel_2_world) for the geometry this scene has: ground plane, table, cubes, box hulls of the robot.
//
// HBM-bound by construction: 8 B are written per pixel (f32 depth, 3 x u8 colour, u8 geom id) and
// the only reads are 1 KB of geom poses per workgroup.  One thread owns 4 horizontally adjacent pixels
// so that depth leaves as one 16-byte store per lane, colour as three dwords and ids as one dword; a
// workgroup of 320 threads covers two image rows per iteration.  Per workgroup the 16 geoms are turned
// into (camera origin, ray basis) in their own frames plus a screen-space bounding rectangle, so a
// pixel group only intersects the geoms whose rectangle it touches.
//
// Ground plane and table are fixed to the world: their image is the same for every env and frame of a
// camera.  mre_render (mre_api.cpp) renders it once (geoms [0, 2), N = 1) and keeps it; a frame then
// starts every pixel from that background -- 2.4 MB shared by all workgroups, served from L2 /
// Infinity Cache -- and casts only the moving geoms [2, 16), whose rectangles cover a few percent of
// the image, so most 4-pixel groups are a copy (1.97 ms per 4096-env frame = 0.64 of HBM peak,
// against 3.7 ms when every pixel casts all geoms).
//
// Shading is a documented approximation of MuJoCo's fixed-function lighting (no specular, no shadows,
// no textures except the ground checker): albedo * (ambient + headlight diffuse * cos(view) + scene
// light diffuse * cos(light)); depth is the distance along the optical axis, like mujoco.Renderer's.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mre_dev.h"

namespace mre {

constexpr int RENDER_THREADS = 320;  // 5 wavefronts = 2 rows of 640 pixels, 4 pixels per thread
constexpr float RENDER_NEAR = 0.01f;

struct GeomView {
  float o[3];      // camera position in the geom frame
  float G[9];      // geom-frame ray direction = G * (x, y, -1), row-major
  float size[3];
  float Lg[3];     // scene light position minus camera position, in the geom frame
  float rgb[3];
  int type;        // 0 plane, > 0 box, -1 unused
  int x0, x1, y0, y1;  // inclusive screen rectangle
};

__global__ __launch_bounds__(RENDER_THREADS) void k_render(RenderArgs a) {
  __shared__ GeomView gv[NG];
  __shared__ int rect[NG][4];
  const int env = blockIdx.x, t = threadIdx.x;
  if (env >= a.N) return;
  if (a.env_mask != nullptr && a.env_mask[env] == 0) return;
  const float cx = 0.5f * (a.width - 1), cy = 0.5f * (a.height - 1), f = a.fy, inv_f = 1.0f / a.fy;
  const float inv_checker = 1.0f / a.checker_size;
  // ---- per-geom view data (thread = geom) and projected corners (thread = geom * 8 + corner)
  if (t < NG) {
    rect[t][0] = a.width; rect[t][1] = -1; rect[t][2] = a.height; rect[t][3] = -1;
    const float* g = a.geoms + ((size_t)env * NG + t) * 16;
    GeomView& v = gv[t];
    float rel[3];
    for (int k = 0; k < 3; k++) rel[k] = a.cam_pos[k] - g[k];
    for (int i = 0; i < 3; i++) {  // o = R' rel ; G = R' Rc
      v.o[i] = g[3 + i] * rel[0] + g[6 + i] * rel[1] + g[9 + i] * rel[2];
      for (int j = 0; j < 3; j++)
        v.G[3 * i + j] = g[3 + i] * a.cam_mat[j] + g[6 + i] * a.cam_mat[3 + j] + g[9 + i] * a.cam_mat[6 + j];
      v.size[i] = g[12 + i];
    }
    {
      float lr[3];
      for (int k = 0; k < 3; k++) lr[k] = a.light_pos[k] - a.cam_pos[k];
      for (int i = 0; i < 3; i++) v.Lg[i] = g[3 + i] * lr[0] + g[6 + i] * lr[1] + g[9 + i] * lr[2];
    }
    v.type = (int)g[15];
    const int pid = (t >= PROP_GEOM0 && t < PROP_GEOM0 + NPROP) ? t - PROP_GEOM0 : -1;  // cube geoms
    for (int k = 0; k < 3; k++)
      v.rgb[k] = (pid >= 0) ? a.prop_rgb[((size_t)env * NPROP + pid) * 3 + k] * (1.0f / 255.0f) : a.geom_rgb[t][k];
  }
  __syncthreads();
  if (t < NG * 8) {
    const int g = t >> 3, c = t & 7;
    const float* q = a.geoms + ((size_t)env * NG + g) * 16;
    const int type = (int)q[15];
    if (type > 0) {
      const float s[3] = {(c & 1) ? q[12] : -q[12], (c & 2) ? q[13] : -q[13], (c & 4) ? q[14] : -q[14]};
      float w[3], cc[3];
      for (int i = 0; i < 3; i++) w[i] = q[i] + q[3 + 3 * i] * s[0] + q[4 + 3 * i] * s[1] + q[5 + 3 * i] * s[2] - a.cam_pos[i];
      for (int j = 0; j < 3; j++) cc[j] = a.cam_mat[j] * w[0] + a.cam_mat[3 + j] * w[1] + a.cam_mat[6 + j] * w[2];
      if (cc[2] > -RENDER_NEAR) {  // corner behind the camera: keep the whole screen
        atomicMin(&rect[g][0], 0); atomicMax(&rect[g][1], a.width - 1);
        atomicMin(&rect[g][2], 0); atomicMax(&rect[g][3], a.height - 1);
      } else {
        const float u = cx + f * cc[0] / (-cc[2]), vv = cy - f * cc[1] / (-cc[2]);
        atomicMin(&rect[g][0], (int)floorf(u) - 1); atomicMax(&rect[g][1], (int)ceilf(u) + 1);
        atomicMin(&rect[g][2], (int)floorf(vv) - 1); atomicMax(&rect[g][3], (int)ceilf(vv) + 1);
      }
    } else if (type == 0 && c == 0) {
      rect[g][0] = 0; rect[g][1] = a.width - 1; rect[g][2] = 0; rect[g][3] = a.height - 1;
    }
  }
  __syncthreads();
  if (t < NG) { gv[t].x0 = rect[t][0]; gv[t].x1 = rect[t][1]; gv[t].y0 = rect[t][2]; gv[t].y1 = rect[t][3]; }
  __syncthreads();

  const int groups_per_row = a.width >> 2;                 // 4-pixel groups per row
  const int rows_per_iter = RENDER_THREADS / groups_per_row;  // 2 for 640-wide images
  const int tr = t / groups_per_row, tg = t - tr * groups_per_row;
  const size_t img = (size_t)env * a.height * a.width;
  for (int row0 = blockIdx.y * rows_per_iter; row0 < a.height; row0 += gridDim.y * rows_per_iter) {
    const int row = row0 + tr;
    if (tr >= rows_per_iter || row >= a.height) continue;
    const int u0 = tg << 2;
    const float y = -(row - cy) * inv_f;
    float best_t[4], cosv[4];
    int best_g[4], best_ax[4];
#pragma unroll
    for (int p = 0; p < 4; p++) { best_t[p] = a.zfar; best_g[p] = 255; best_ax[p] = 0; cosv[p] = 0.f; }
    // start from the cached image of the static geoms (2.4 MB, shared by all envs: stays in L2 /
    // Infinity Cache); a pixel keeps it unless a moving geom is nearer.  254 = "background pixel".
    const bool use_bg = a.bg_depth != nullptr;
    const size_t bpix = (size_t)row * a.width + u0;
    uint32_t bg_rgbw[3] = {0u, 0u, 0u}, bg_segw = 0u;
    if (use_bg) {
      const float4 bd = *reinterpret_cast<const float4*>(a.bg_depth + bpix);
      best_t[0] = bd.x; best_t[1] = bd.y; best_t[2] = bd.z; best_t[3] = bd.w;
      const uint32_t* br = reinterpret_cast<const uint32_t*>(a.bg_rgb + bpix * 3);
      bg_rgbw[0] = br[0]; bg_rgbw[1] = br[1]; bg_rgbw[2] = br[2];
      bg_segw = *reinterpret_cast<const uint32_t*>(a.bg_seg + bpix);
#pragma unroll
      for (int p = 0; p < 4; p++) best_g[p] = 254;
    }
    // geoms whose screen rectangle reaches this iteration's rows (wave-uniform bit mask: every
    // wave evaluates the 16 tests in its first 16 lanes), then only those are visited
    const int wl = t & 63;
    const int gq = wl < NG ? wl : 0;
    const bool rows_hit = wl >= a.g0 && wl < a.g1 && gv[gq].type >= 0 && gv[gq].y0 <= row0 + rows_per_iter - 1 && gv[gq].y1 >= row0;
    unsigned long long todo = __ballot(rows_hit);
    while (todo != 0ull) {
      const int g = __builtin_ctzll(todo);
      todo &= todo - 1ull;
      const GeomView& v = gv[g];
      if (row < v.y0 || row > v.y1 || u0 + 3 < v.x0 || u0 > v.x1) continue;
      // ray direction in the geom frame is affine in the pixel column: d = G[:,0] x + e
      const float e0 = v.G[1] * y - v.G[2], e1 = v.G[4] * y - v.G[5], e2 = v.G[7] * y - v.G[8];
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const float x = (u0 + p - cx) * inv_f;
        const float d0 = v.G[0] * x + e0, d1 = v.G[3] * x + e1, d2 = v.G[6] * x + e2;
        float th, cs;
        int ax;
        if (v.type == 0) {  // plane z = 0 of the geom frame, visible from above
          if (!(d2 < 0.f)) continue;
          th = -v.o[2] * __builtin_amdgcn_rcpf(d2);
          ax = 2; cs = d2;
        } else {
          // the ray enters a box through a face the camera is in front of: for each axis whose slab
          // the camera is outside of (wave-uniform per geom) intersect that one face plane and keep
          // the hit if it lies within the face.  (Seen from above the table is ONE such axis.)
          const float dd3[3] = {d0, d1, d2};
          bool found = false;
          th = 0.f; ax = 0; cs = 0.f;
#pragma unroll
          for (int k = 0; k < 3; k++) {
            if (fabsf(v.o[k]) <= v.size[k]) continue;
            const float face = v.o[k] > 0.f ? v.size[k] : -v.size[k];
            const float tk = (face - v.o[k]) * __builtin_amdgcn_rcpf(dd3[k]);
            const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
            const float p1 = v.o[k1] + tk * dd3[k1], p2 = v.o[k2] + tk * dd3[k2];
            const bool ok = tk > RENDER_NEAR && fabsf(p1) <= v.size[k1] && fabsf(p2) <= v.size[k2];
            if (ok && !found) { found = true; th = tk; ax = k; cs = dd3[k]; }
          }
          if (!found) continue;
        }
        if (th > RENDER_NEAR && th < best_t[p]) { best_t[p] = th; best_g[p] = g; best_ax[p] = ax; cosv[p] = cs; }
      }
    }
    // ---- shade the winning surface and store.  With n the entry face's normal (geom axis ax, turned
    // towards the camera) and ds the ray direction's component along that axis (kept from the hit):
    //   cos(view)  = |ds| / |d|
    //   n.(L - hit) = sgn * (Lg[ax] - t ds),   |L - hit|^2 = |L-c|^2 - 2 t (L-c).d + t^2 |d|^2
    float lc[3], l2 = 0.f;
    for (int j = 0; j < 3; j++) {
      lc[j] = 0.f;
      for (int r = 0; r < 3; r++) lc[j] += a.cam_mat[3 * r + j] * (a.light_pos[r] - a.cam_pos[r]);
      l2 += (a.light_pos[j] - a.cam_pos[j]) * (a.light_pos[j] - a.cam_pos[j]);
    }
    uint32_t rgbw[3] = {0u, 0u, 0u};
    uint32_t segw = 0u;
    bool any_fg = false;
#pragma unroll
    for (int p = 0; p < 4; p++) any_fg = any_fg || best_g[p] < 254;
    if (use_bg && !any_fg) {  // the common case: all four pixels are background
      rgbw[0] = bg_rgbw[0]; rgbw[1] = bg_rgbw[1]; rgbw[2] = bg_rgbw[2];
      segw = bg_segw;
    } else {
#pragma unroll
    for (int p = 0; p < 4; p++) {
      if (best_g[p] == 254) {  // background pixel next to a moving geom: take its cached bytes
        for (int i = 0; i < 3; i++) {
          const int pos = 3 * p + i;
          rgbw[pos >> 2] |= ((bg_rgbw[pos >> 2] >> (8 * (pos & 3))) & 0xFFu) << (8 * (pos & 3));
        }
        segw |= ((bg_segw >> (8 * p)) & 0xFFu) << (8 * p);
        continue;
      }
      float col[3] = {0.4f, 0.6f, 0.8f};  // skybox tint where nothing is hit
      if (best_g[p] != 255) {
        const GeomView& v = gv[best_g[p]];
        const float x = (u0 + p - cx) * inv_f;
        const float dd = x * x + y * y + 1.0f, th = best_t[p], ds = cosv[p];
        const float cview = fabsf(ds) * rsqrtf(dd);
        const float sgn = ds > 0.f ? -1.f : 1.f;
        const float ldd = lc[0] * x + lc[1] * y - lc[2];
        const float lv2 = l2 - 2.f * th * ldd + th * th * dd;
        const float cl = fmaxf(0.f, sgn * (v.Lg[best_ax[p]] - th * ds)) * rsqrtf(lv2);
        const float inten = a.ambient + a.head_diffuse * cview + a.light_diffuse * cl;
        float alb[3] = {v.rgb[0], v.rgb[1], v.rgb[2]};
        if (v.type == 0) {  // ground checker in the plane's own xy
          const float px = v.o[0] + th * (v.G[0] * x + v.G[1] * y - v.G[2]);
          const float py = v.o[1] + th * (v.G[3] * x + v.G[4] * y - v.G[5]);
          const int par = ((int)floorf(px * inv_checker) + (int)floorf(py * inv_checker)) & 1;
          for (int i = 0; i < 3; i++) alb[i] = a.checker[par][i];
        }
        for (int i = 0; i < 3; i++) col[i] = fminf(1.f, alb[i] * inten);
      }
      for (int i = 0; i < 3; i++) {
        const uint32_t byte = (uint32_t)(col[i] * 255.f + 0.5f);
        const int pos = 3 * p + i;  // byte index within the thread's 12 colour bytes
        rgbw[pos >> 2] |= byte << (8 * (pos & 3));
      }
      segw |= (uint32_t)(best_g[p] & 0xFF) << (8 * p);
    }
    }
    const size_t pix = img + (size_t)row * a.width + u0;
    if (a.depth != nullptr)
      *reinterpret_cast<float4*>(a.depth + pix) = make_float4(best_t[0], best_t[1], best_t[2], best_t[3]);
    if (a.seg != nullptr) *reinterpret_cast<uint32_t*>(a.seg + pix) = segw;
    if (a.rgb != nullptr) {
      uint32_t* o = reinterpret_cast<uint32_t*>(a.rgb + pix * 3);
      o[0] = rgbw[0]; o[1] = rgbw[1]; o[2] = rgbw[2];
    }
  }
}

}  // namespace mre

extern "C" void mre_launch_render(const mre::RenderArgs* args, int row_groups, hipStream_t stream) {
  hipLaunchKernelGGL(mre::k_render, dim3(args->N, row_groups), dim3(mre::RENDER_THREADS), 0, stream, *args);
}
