// Host-side plumbing shared by all entry points of libdiner_hip.so.
#include <stdarg.h>
#include "common.hpp"

namespace diner {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int make_scene_dev(const DinerScene* s, SceneDev* out) {
  DINER_CHECK_ARG(s != nullptr, "scene is null");
  DINER_CHECK_ARG(s->nv >= 1 && s->nv <= kMaxViews, "scene: nv=%d outside [1,%d]", s->nv, kMaxViews);
  DINER_CHECK_ARG(s->poses && s->focal && s->c, "scene: poses/focal/c (host arrays) missing");
  DINER_CHECK_ARG(s->img_w > 0 && s->img_h > 0, "scene: image_shape must be positive");
  memset(out, 0, sizeof(*out));
  out->latent_cl = s->latent_cl;
  out->depth = s->depth;
  out->depth_std = s->depth_std;
  out->normals = s->normals;
  out->std_pad_scale = s->std_pad_scale;
  for (int v = 0; v < s->nv; ++v) {
    const float* P = s->poses + 16 * v;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) out->R[v][3 * i + j] = P[4 * i + j];
      out->t[v][i] = P[4 * i + 3];
    }
    out->focal[v][0] = s->focal[2 * v];
    out->focal[v][1] = s->focal[2 * v + 1];
    out->c[v][0] = s->c[2 * v];
    out->c[v][1] = s->c[2 * v + 1];
  }
  out->img_w = s->img_w;
  out->img_h = s->img_h;
  out->feature_padding = s->feature_padding;
  out->nv = s->nv;
  out->C = s->C;
  out->Hf = s->Hf;
  out->Wf = s->Wf;
  out->Hs = s->Hs;
  out->Ws = s->Ws;
  return 0;
}

}  // namespace diner

extern "C" int diner_abi_version(void) { return DINER_ABI_VERSION; }
extern "C" const char* diner_last_error(void) { return diner::g_err; }
