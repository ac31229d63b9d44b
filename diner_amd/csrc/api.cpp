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
  DINER_CHECK_ARG(s->poses_host && s->focal_host && s->c_host, "scene: poses_host/focal_host/c_host (host arrays) missing");
  DINER_CHECK_ARG(s->img_w > 0 && s->img_h > 0, "scene: image_shape must be positive");
  memset(out, 0, sizeof(*out));
  out->latent_cl = s->latent_cl;
  out->depth = s->depth;
  out->depth_std = s->depth_std;
  out->normals = s->normals;
  out->std_pad_scale = s->std_pad_scale;
  for (int v = 0; v < s->nv; ++v) {
    const float* P = s->poses_host + 16 * v;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) out->R[v][3 * i + j] = P[4 * i + j];
      out->t[v][i] = P[4 * i + 3];
    }
    out->focal[v][0] = s->focal_host[2 * v];
    out->focal[v][1] = s->focal_host[2 * v + 1];
    out->c[v][0] = s->c_host[2 * v];
    out->c[v][1] = s->c_host[2 * v + 1];
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

int check_mlp_config(const DinerMlpParams* p, const char* who, bool poscode) {
  DINER_CHECK_ARG(p && p->lin_in_w && p->lin_in_b && p->lin_out_w && p->lin_out_b && p->fc0_w && p->fc0_b && p->fc1_w &&
                      p->fc1_b && p->lin_z_w && p->lin_z_b, "%s: parameter pointers missing", who);
  if (p->d_in != 55 || p->d_latent != 512 || p->d_hidden != 512 || p->d_out != 4 || p->n_blocks != 5 ||
      p->combine_layer != 3) {
    set_error("%s: unsupported ResnetFC configuration d_in=%d d_latent=%d d_hidden=%d d_out=%d n_blocks=%d "
              "combine_layer=%d (built for 55/512/512/4/5/3, configs/train_dtu.yaml:44-50)",
              who, p->d_in, p->d_latent, p->d_hidden, p->d_out, p->n_blocks, p->combine_layer);
    return DINER_E_UNSUPPORTED;
  }
  if (poscode) {
    // d_in = 3 * (2 F + 1) + (2 F + 1) + 3 = 55 fixes F = 6 with the input included (pixelnerf.py:15-18)
    if (p->num_freqs != 6 || !p->include_input) {
      set_error("%s: unsupported positional encoding num_freqs=%d include_input=%d (built for 6 / 1, "
                "configs/train_dtu.yaml:39-43)", who, p->num_freqs, p->include_input);
      return DINER_E_UNSUPPORTED;
    }
    // the in-kernel sine is accurate for arguments below ~2^13: |x_c| * freq_factor * 32 with |x_c| up to a few units
    if (!(p->freq_factor > 0.0f && p->freq_factor <= 64.0f)) {
      set_error("%s: unsupported positional encoding freq_factor=%g (supported: (0, 64]; 6.28 in the shipped configs)",
                who, (double)p->freq_factor);
      return DINER_E_UNSUPPORTED;
    }
  }
  return 0;
}

}  // namespace diner

extern "C" int diner_abi_version(void) { return DINER_ABI_VERSION; }
extern "C" const char* diner_last_error(void) { return diner::g_err; }
