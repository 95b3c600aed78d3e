// stacks.hip -- whole-stack entry points of the C ABI (include/mvq.h, "whole stacks"): mvq_encoder_fwd_f32, mvq_decoder_fwd_f32,
// mvq_decoder_fwd_saving_f32 / mvq_decoder_bwd_input_f32.
//
// They replace `A_ENC(a)` / `T_ENC(t)` / `T_DEC(z)` and the input-gradient of `T_DEC` under `.backward()`
// (Training/compare_dacvsproposal_5.py:294,296,322,393) with ONE C call each.  A stack handle is made once per weight load from the
// upstream state-dict tensors (weight_g / weight_v / bias / alpha, in the order mvq_*_param_info reports): the weight-norm fold and
// the packed images live in a caller-provided device blob.  A call walks the launch plan that used to exist only in the Python
// mirror (dual outputs so that no wide layer evaluates a Snake while staging, zero-padded rows, packed / virtually packed
// latent-rate rows at throughput batch sizes, one fused launch per narrow ResidualUnit) over the per-layer entry points of this
// library, with its intermediates in a caller-provided workspace laid out by a deterministic first-fit arena: no allocation,
// no synchronisation, capturable into a hipGraph.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <string>
#include <vector>
#include "../../include/mvq.h"

namespace mvq { void set_last_error(const char* msg); }        // api.hip: the text mvq_last_error() returns

namespace {

int sfail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    mvq::set_last_error(buf);
    return code;
}

constexpr size_t ALIGN = 256;
inline size_t up(size_t n) { return (n + ALIGN - 1) / ALIGN * ALIGN; }
inline int ceil4(int t) { return (t + 3) / 4 * 4; }
inline int conv_len(int tin, int ks, int stride, int dil, int pad)
{
    const int span = tin + 2 * pad - dil * (ks - 1) - 1;
    return span < 0 ? 0 : span / stride + 1;
}

// ---- workspace arena: first fit over a free list, deterministic for a given call sequence.  `base == nullptr`: planning run
// (only the peak is wanted); otherwise the same sequence hands out real addresses.
struct Arena {
    char* base = nullptr;
    size_t cap = 0, peak = 0;
    std::vector<std::pair<size_t, size_t>> free_;          // (offset, bytes), sorted by offset, coalesced
    std::map<size_t, size_t> live;                         // offset -> bytes
    size_t top = 0;
    bool overflow = false;
    float* alloc(size_t floats)
    {
        const size_t n = up(floats * sizeof(float) + 16);
        for (size_t i = 0; i < free_.size(); ++i)
            if (free_[i].second >= n) {
                const size_t off = free_[i].first;
                if (free_[i].second == n) free_.erase(free_.begin() + i);
                else { free_[i].first += n; free_[i].second -= n; }
                live[off] = n;
                return at(off);
            }
        const size_t off = top;
        top += n;
        if (top > peak) peak = top;
        live[off] = n;
        return at(off);
    }
    float* at(size_t off)
    {
        if (base && off + live[off] > cap) overflow = true;
        return base ? reinterpret_cast<float*>(base + off) : reinterpret_cast<float*>(ALIGN + off);      // planning: fake, never dereferenced
    }
    void release(float* p)
    {
        if (!p) return;
        const size_t off = base ? (size_t)(reinterpret_cast<char*>(p) - base) : (size_t)(reinterpret_cast<size_t>(p) - ALIGN);
        auto it = live.find(off);
        if (it == live.end()) return;
        const size_t n = it->second;
        live.erase(it);
        if (off + n == top) {                                  // shrink the bump pointer, swallowing a free block that now touches it
            top = off;
            while (!free_.empty() && free_.back().first + free_.back().second == top) { top = free_.back().first; free_.pop_back(); }
            return;
        }
        size_t i = 0;
        while (i < free_.size() && free_[i].first < off) ++i;
        free_.insert(free_.begin() + i, {off, n});
        if (i + 1 < free_.size() && free_[i].first + free_[i].second == free_[i + 1].first) { free_[i].second += free_[i + 1].second; free_.erase(free_.begin() + i + 1); }
        if (i > 0 && free_[i - 1].first + free_[i - 1].second == free_[i].first) { free_[i - 1].second += free_[i].second; free_.erase(free_.begin() + i); }
    }
};

struct Conv {            // one weight-normed conv (or transposed conv) of a stack
    int cin = 0, cout = 0, ks = 0, stride = 1, dil = 1, pad = 0, opad = 0;
    bool transposed = false;
    int p_g = -1, p_v = -1, p_b = -1;          // parameter indices (weight_g, weight_v, bias)
    const float* wp = nullptr;                 // packed forward image
    const float* wd = nullptr;                 // packed input-gradient image (decoder only)
    const float* bias = nullptr;
    size_t w_floats() const { return (size_t)cin * cout * ks; }
};
struct Unit {            // dac ResidualUnit: snake_a, conv7(dil), snake_b, conv1
    Conv c7, c1;
    int p_aa = -1, p_ab = -1;
    const float* alpha_a = nullptr; const float* alpha_b = nullptr;
};
struct Block {           // EncoderBlock: 3 units, snake, strided conv.  DecoderBlock: snake, transposed conv, 3 units
    Unit ru[3];
    Conv res;            // the resampling conv
    int p_alpha = -1;
    const float* alpha = nullptr;
    int width = 0;       // channels of the units
};

}  // namespace

struct mvq_stack {
    bool decoder = false;
    std::vector<std::string> names;
    std::vector<std::vector<int>> shapes;
    Conv first, last;
    std::vector<Block> blocks;
    int p_alpha_last = -1;
    const float* alpha_last = nullptr;         // encoder: Snake before the k3 conv; decoder: Snake before the output conv
    int out_padding = 0;
    // plan switches (the Python mirror's class attributes): batch thresholds of the packed latent-rate forms
    int vpack_seg = 10, vpack_min_batch = 32, pack_seg = 8, pack_min_batch = 32, presnaked_min_c = 65;
};

namespace {

int add_param(mvq_stack& s, const std::string& name, std::vector<int> shape)
{
    s.names.push_back(name); s.shapes.push_back(shape);
    return (int)s.names.size() - 1;
}
void add_conv_params(mvq_stack& s, Conv& c, const std::string& prefix)
{
    const int rows = c.transposed ? c.cin : c.cout, cols = c.transposed ? c.cout : c.cin;
    c.p_g = add_param(s, prefix + ".weight_g", {rows, 1, 1});
    c.p_v = add_param(s, prefix + ".weight_v", {rows, cols, c.ks});
    c.p_b = add_param(s, prefix + ".bias", {c.cout});
}
void add_unit(mvq_stack& s, Unit& u, const std::string& prefix, int c, int dil)
{
    u.c7.cin = u.c7.cout = c; u.c7.ks = 7; u.c7.dil = dil; u.c7.pad = 3 * dil;
    u.c1.cin = u.c1.cout = c; u.c1.ks = 1;
    u.p_aa = add_param(s, prefix + ".block.0.alpha", {1, c, 1});
    add_conv_params(s, u.c7, prefix + ".block.1");
    u.p_ab = add_param(s, prefix + ".block.2.alpha", {1, c, 1});
    add_conv_params(s, u.c1, prefix + ".block.3");
}

// upstream dac Encoder: block.0 conv(1 -> d, k7), block.i EncoderBlock(d_i, stride_i), block.n+1 Snake, block.n+2 conv(d -> d_latent, k3)
int build_encoder(mvq_stack& s, const mvq_encoder_desc& d)
{
    if (d.d_model <= 0 || d.n_strides <= 0 || d.n_strides > 8 || d.d_latent <= 0) return MVQ_EINVAL;
    s.decoder = false;
    int dm = d.d_model;
    s.first.cin = 1; s.first.cout = dm; s.first.ks = 7; s.first.pad = 3;
    add_conv_params(s, s.first, "block.0");
    for (int i = 0; i < d.n_strides; ++i) {
        const int st = d.strides[i];
        if (st <= 0) return MVQ_EINVAL;
        dm *= 2;
        Block b;
        b.width = dm / 2;
        const std::string p = "block." + std::to_string(i + 1) + ".block.";
        const int dils[3] = {1, 3, 9};
        for (int j = 0; j < 3; ++j) add_unit(s, b.ru[j], p + std::to_string(j), dm / 2, dils[j]);
        b.p_alpha = add_param(s, p + "3.alpha", {1, dm / 2, 1});
        b.res.cin = dm / 2; b.res.cout = dm; b.res.ks = 2 * st; b.res.stride = st; b.res.pad = (st + 1) / 2;
        add_conv_params(s, b.res, p + "4");
        s.blocks.push_back(b);
    }
    s.p_alpha_last = add_param(s, "block." + std::to_string(d.n_strides + 1) + ".alpha", {1, dm, 1});
    s.last.cin = dm; s.last.cout = d.d_latent; s.last.ks = 3; s.last.pad = 1;
    add_conv_params(s, s.last, "block." + std::to_string(d.n_strides + 2));
    return MVQ_OK;
}

// upstream dac Decoder: model.0 conv(in -> ch, k7), model.i DecoderBlock(ch / 2^(i-1) -> ch / 2^i, stride), Snake, conv(-> d_out, k7), Tanh
int build_decoder(mvq_stack& s, const mvq_decoder_desc& d)
{
    if (d.input_channel <= 0 || d.channels <= 0 || d.n_rates <= 0 || d.n_rates > 8 || d.d_out <= 0) return MVQ_EINVAL;
    s.decoder = true;
    s.out_padding = d.output_padding ? 1 : 0;
    s.first.cin = d.input_channel; s.first.cout = d.channels; s.first.ks = 7; s.first.pad = 3;
    add_conv_params(s, s.first, "model.0");
    int out = d.channels;
    for (int i = 0; i < d.n_rates; ++i) {
        const int st = d.rates[i];
        if (st <= 0) return MVQ_EINVAL;
        const int inp = d.channels >> i;
        out = d.channels >> (i + 1);
        if (out <= 0) return MVQ_EINVAL;
        Block b;
        b.width = out;
        const std::string p = "model." + std::to_string(i + 1) + ".block.";
        b.p_alpha = add_param(s, p + "0.alpha", {1, inp, 1});
        b.res.transposed = true; b.res.cin = inp; b.res.cout = out; b.res.ks = 2 * st; b.res.stride = st; b.res.pad = (st + 1) / 2;
        b.res.opad = d.output_padding ? st % 2 : 0;
        add_conv_params(s, b.res, p + "1");
        const int dils[3] = {1, 3, 9};
        for (int j = 0; j < 3; ++j) add_unit(s, b.ru[j], p + std::to_string(j + 2), out, dils[j]);
        s.blocks.push_back(b);
    }
    s.p_alpha_last = add_param(s, "model." + std::to_string(d.n_rates + 1) + ".alpha", {1, out, 1});
    s.last.cin = out; s.last.cout = d.d_out; s.last.ks = 7; s.last.pad = 3;
    add_conv_params(s, s.last, "model." + std::to_string(d.n_rates + 2));
    return MVQ_OK;
}

size_t packed_floats(const Conv& c)
{
    return c.transposed ? mvq_conv_transpose1d_packed_floats(c.cin, c.cout, c.stride) : mvq_conv1d_packed_floats(c.cin, c.cout, c.ks);
}
size_t dgrad_floats(const Conv& c) { return mvq_conv1d_dgrad_packed_floats(c.cin, c.cout, c.ks); }

template <class F> void for_each_conv(mvq_stack& s, F f)
{
    f(s.first);
    for (auto& b : s.blocks) {
        if (s.decoder) f(b.res);
        for (auto& u : b.ru) { f(u.c7); f(u.c1); }
        if (!s.decoder) f(b.res);
    }
    f(s.last);
}

size_t weights_bytes(mvq_stack& s)
{
    size_t n = 0, wmax = 0, small = 0;
    for_each_conv(s, [&](Conv& c) {
        n += up(packed_floats(c) * 4);
        if (s.decoder) n += up(dgrad_floats(c) * 4);
        small += up((size_t)c.cout * 4);
        if (c.w_floats() > wmax) wmax = c.w_floats();
    });
    for (auto& b : s.blocks) {
        small += up((size_t)(s.decoder ? b.res.cin : b.width) * 4);
        small += 3 * 2 * up((size_t)b.width * 4);
    }
    small += up((size_t)s.last.cin * 4);
    return n + small + up(wmax * 4);                         // + one folded-weight staging area
}

// fold weight norm, pack (forward and, for a decoder, input-gradient images), copy biases and alphas: everything the plan reads
// afterwards lives in `blob`
int prepare_weights(mvq_stack& s, const float* const* params, void* blob, size_t blob_bytes, void* stream)
{
    if (weights_bytes(s) > blob_bytes) return sfail(MVQ_EINVAL, "stack_create: weights blob too small (%zu < %zu bytes)", blob_bytes, weights_bytes(s));
    for (size_t i = 0; i < s.names.size(); ++i)
        if (!params[i]) return sfail(MVQ_EINVAL, "stack_create: parameter %zu (%s) is null", i, s.names[i].c_str());
    char* p = reinterpret_cast<char*>(blob);
    auto take = [&](size_t bytes) { char* q = p; p += up(bytes); return reinterpret_cast<float*>(q); };
    size_t wmax = 0;
    for_each_conv(s, [&](Conv& c) { if (c.w_floats() > wmax) wmax = c.w_floats(); });
    float* wtmp = take(wmax * 4);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc = MVQ_OK;
    auto copy = [&](int idx, size_t n) -> const float* {
        float* d = take(n * 4);
        if (hipMemcpyAsync(d, params[idx], n * 4, hipMemcpyDeviceToDevice, st) != hipSuccess && rc == MVQ_OK) rc = sfail(MVQ_EHIP, "stack_create: copy of %s failed", s.names[idx].c_str());
        return d;
    };
    for_each_conv(s, [&](Conv& c) {
        if (rc != MVQ_OK) return;
        const int rows = c.transposed ? c.cin : c.cout;
        rc = mvq_weight_norm_f32(params[c.p_v], params[c.p_g], wtmp, rows, (int)(c.w_floats() / rows), stream);
        if (rc != MVQ_OK) return;
        float* wp = take(packed_floats(c) * 4);
        rc = c.transposed ? mvq_conv_transpose1d_pack_f32(wtmp, wp, c.cin, c.cout, c.stride, stream) : mvq_conv1d_pack_f32(wtmp, wp, c.cin, c.cout, c.ks, stream);
        c.wp = wp;
        if (rc == MVQ_OK && s.decoder) {
            float* wd = take(dgrad_floats(c) * 4);
            rc = c.transposed ? mvq_conv_transpose1d_pack_dgrad_f32(wtmp, wd, c.cin, c.cout, c.ks, stream) : mvq_conv1d_pack_dgrad_f32(wtmp, wd, c.cin, c.cout, c.ks, stream);
            c.wd = wd;
        }
        c.bias = copy(c.p_b, c.cout);
    });
    if (rc != MVQ_OK) return rc;
    for (auto& b : s.blocks) {
        b.alpha = copy(b.p_alpha, s.decoder ? b.res.cin : b.width);
        for (auto& u : b.ru) { u.alpha_a = copy(u.p_aa, b.width); u.alpha_b = copy(u.p_ab, b.width); }
    }
    s.alpha_last = copy(s.p_alpha_last, s.last.cin);
    return rc;
}

// ---- plan execution ------------------------------------------------------------------------------------------------------------
struct Run {
    Arena ar;
    bool dry = true;
    void* stream = nullptr;
    int rc = MVQ_OK;
    bool ok() const { return rc == MVQ_OK; }
    void call(int r) { if (rc == MVQ_OK && r != MVQ_OK) rc = r; }
};

struct Act { float* p = nullptr; int c = 0, rows = 0; };        // [B, c, rows] activation in the workspace

// one ResidualUnit through mvq_residual_unit_padded_f32 (one fused launch for C in {64, 96, 128}; otherwise the two launches with
// the intermediate in the arena)
void run_unit(Run& r, const Unit& u, int B, int C, int T, const float* x, const float* x_snaked, const float* alpha_next, float* y,
              float* y2, const float* alpha2, int tvalid)
{
    const size_t sc = mvq_residual_unit_scratch_floats(B, C, T, u.c7.dil);
    float* scratch = sc ? r.ar.alloc(sc) : nullptr;
    if (!r.dry)
        r.call(mvq_residual_unit_padded_f32(x, x_snaked, u.c7.wp, u.c7.bias, u.alpha_a, u.alpha_b, u.c1.wp, u.c1.bias, alpha_next, y, y2, alpha2,
                                            scratch, B, C, T, u.c7.dil, tvalid, r.stream));
    r.ar.release(scratch);
}

// the three units of a block.  wide (C >= presnaked_min_c): every unit takes its input Snake from its producer's dual output
// (xs) and emits the next unit's; the last one applies `alpha_after` (the Snake in front of whatever follows the units) in place.
// Returns the output activation (x and xs are released).
float* run_units(Run& r, const mvq_stack& s, const Block& b, int B, int T, float* x, float* xs, const float* alpha_after, int tvalid)
{
    const int C = b.width;
    const size_t n = (size_t)B * C * T;
    const bool wide = C >= s.presnaked_min_c;
    for (int j = 0; j < 3; ++j) {
        const bool lastu = j == 2;
        float* y = r.ar.alloc(n);
        float* y2 = (wide && !lastu) ? r.ar.alloc(n) : nullptr;
        run_unit(r, b.ru[j], B, C, T, x, wide ? xs : nullptr, lastu ? alpha_after : nullptr, y, y2, y2 ? b.ru[j + 1].alpha_a : nullptr, tvalid);
        r.ar.release(x); r.ar.release(xs);
        x = y; xs = y2;
    }
    return x;
}

// geometry of mvq_conv1d_vpacked_f32 (the Python mirror's ops.vpacked_geometry)
bool vpacked_geometry(int tin_rows, int tin_valid, int ks, int stride, int dil, int pad, int follow_pad, int* tout, int* tout_rows)
{
    *tout = conv_len(tin_valid, ks, stride, dil, pad);
    if (tin_rows % 4 || *tout <= 0) return false;
    const int overhang = (*tout - 1) * stride - pad + dil * (ks - 1) - (tin_valid - 1);
    int gap = pad > overhang ? pad : overhang;
    if (gap < 0) gap = 0;
    int a = *tout + follow_pad, b2 = (tin_valid + gap + stride - 1) / stride, c = (tin_rows + stride - 1) / stride;
    int m = a > b2 ? a : b2; m = m > c ? m : c;
    *tout_rows = ceil4(m);
    return true;
}

int encoder_len(const mvq_stack& s, int t)
{
    t = conv_len(t, s.first.ks, 1, 1, s.first.pad);
    for (auto& b : s.blocks) t = conv_len(t, b.res.ks, b.res.stride, 1, b.res.pad);
    return conv_len(t, s.last.ks, 1, 1, s.last.pad);
}

// x[B, 1, T] -> z[B, d_latent, Tl].  use_vpack: the last strided conv and the k3 conv on virtually packed rows (throughput batches)
void encoder_plan(Run& r, const mvq_stack& s, const float* x, float* z, int B, int T, bool use_vpack)
{
    const int nb = (int)s.blocks.size();
    int t = conv_len(T, 7, 1, 1, 3);
    const bool wide0 = s.blocks[0].width >= s.presnaked_min_c;
    float* h = r.ar.alloc((size_t)B * s.first.cout * t);
    float* hs = wide0 ? r.ar.alloc((size_t)B * s.first.cout * t) : nullptr;
    if (!r.dry)
        r.call(mvq_conv1d_padded_f32(x, s.first.wp, s.first.bias, nullptr, nullptr, nullptr, h, hs, hs ? s.blocks[0].ru[0].alpha_a : nullptr, B, 1, T,
                                     s.first.cout, 7, 1, 1, 3, MVQ_ACT_NONE, 0, r.stream));
    for (int i = 0; i < nb && r.ok(); ++i) {
        const Block& b = s.blocks[i];
        const bool last = i == nb - 1;
        h = run_units(r, s, b, B, t, h, hs, b.alpha, 0);
        hs = nullptr;
        const Conv& d = b.res;
        const int tout = conv_len(t, d.ks, d.stride, 1, d.pad);
        const float* a_out = last ? s.alpha_last : nullptr;
        int vt = 0, vrows = 0;
        if (last && use_vpack && vpacked_geometry(t, t, d.ks, d.stride, 1, d.pad, s.last.pad, &vt, &vrows) && d.cin % 32 == 0 && d.cout % 128 == 0) {
            // latent-rate layers on virtually packed rows: y[B, C, rows] carries vt valid columns + a zero tail; the k3 conv maps
            // rows -> rows (its padding is that zero tail); the tail is cut off by the final strided copy
            float* y = r.ar.alloc((size_t)B * d.cout * vrows);
            if (!r.dry)
                r.call(mvq_conv1d_vpacked_f32(h, d.wp, d.bias, nullptr, a_out, y, nullptr, nullptr, B, d.cin, t, t, d.cout, d.ks, d.stride, 1, d.pad,
                                              MVQ_ACT_NONE, s.vpack_seg, d.stride * vrows, vrows, r.stream));
            r.ar.release(h);
            float* y3 = r.ar.alloc((size_t)B * s.last.cout * vrows);
            if (!r.dry) {
                r.call(mvq_conv1d_vpacked_f32(y, s.last.wp, s.last.bias, nullptr, nullptr, y3, nullptr, nullptr, B, s.last.cin, vrows, vt, s.last.cout, 3, 1, 1,
                                              1, MVQ_ACT_NONE, s.vpack_seg, vrows, vrows, r.stream));
                r.call(mvq_copy3d_f32(y3, (size_t)s.last.cout * vrows, (size_t)vrows, z, (size_t)s.last.cout * vt, (size_t)vt, B, s.last.cout, vt, r.stream));
            }
            r.ar.release(y); r.ar.release(y3);
            return;
        }
        const bool wide_next = !last && s.blocks[i + 1].width >= s.presnaked_min_c;
        float* y = r.ar.alloc((size_t)B * d.cout * tout);
        float* y2 = wide_next ? r.ar.alloc((size_t)B * d.cout * tout) : nullptr;
        if (!r.dry)
            r.call(mvq_conv1d_padded_f32(h, d.wp, d.bias, nullptr, nullptr, a_out, y, y2, y2 ? s.blocks[i + 1].ru[0].alpha_a : nullptr, B, d.cin, t, d.cout,
                                         d.ks, d.stride, 1, d.pad, MVQ_ACT_NONE, 0, r.stream));
        r.ar.release(h);
        h = y; hs = y2; t = tout;
    }
    if (!r.ok()) return;
    if (!r.dry)
        r.call(mvq_conv1d_padded_f32(h, s.last.wp, s.last.bias, nullptr, nullptr, nullptr, z, nullptr, nullptr, B, s.last.cin, t, s.last.cout, 3, 1, 1, 1,
                                     MVQ_ACT_NONE, 0, r.stream));
    r.ar.release(h);
}

int convtr_len(const Conv& c, int tin) { return (tin - 1) * c.stride - 2 * c.pad + 2 * c.stride + c.opad; }
int decoder_len(const mvq_stack& s, int t)
{
    if (t <= 0) return 0;
    for (auto& b : s.blocks) t = convtr_len(b.res, t);
    return t;
}

// one DecoderBlock on (possibly zero-padded) rows: x[B, cin, rows_in] carries t_in true columns and already the block's leading
// Snake.  Returns the output; *t_out = its true length, *rows_out its row length.
float* decoder_block(Run& r, const mvq_stack& s, const Block& b, int B, float* x, int rows_in, int t_in, const float* alpha_after, int* t_out,
                     int* rows_out)
{
    const Conv& up_ = b.res;
    const int tn = convtr_len(up_, t_in), tnat = convtr_len(up_, rows_in), tp = ceil4(tn);
    const bool can_pad = tp != tn && tp <= tnat - up_.opad + up_.pad && up_.cin % 32 == 0 && up_.cout % 32 == 0;
    int rows = can_pad ? tp : tn;
    const int tv = can_pad ? tn : 0;
    int arg_rows = (rows == tnat && tv == 0) ? 0 : rows;
    const bool wide = b.width >= s.presnaked_min_c;
    const size_t n = (size_t)B * up_.cout * rows;
    float* h = r.ar.alloc(n);
    float* hs = wide ? r.ar.alloc(n) : nullptr;
    if (!r.dry)
        r.call(mvq_conv_transpose1d_op_f32(x, up_.wp, up_.bias, nullptr, nullptr, h, hs, hs ? b.ru[0].alpha_a : nullptr, B, up_.cin, rows_in, up_.cout,
                                           up_.stride, up_.pad, up_.opad, arg_rows, tv, r.stream));
    r.ar.release(x);
    *t_out = tn; *rows_out = rows;
    return run_units(r, s, b, B, rows, h, hs, alpha_after, tv);
}

bool decoder_use_packed(const mvq_stack& s, int B, int t)
{
    const Conv& up_ = s.blocks[0].res;
    return B >= s.pack_min_batch && t > 0 && t <= 1024 && s.first.cin % 32 == 0 && s.first.cout % 128 == 0 && up_.opad == 0 && up_.cin % 32 == 0 &&
           (up_.cout * up_.stride) % 128 == 0 && ((t - 1) * up_.stride - 2 * up_.pad + 2 * up_.stride) % 4 == 0;
}

// z[B, C, t] -> y[B, d_out, Tout]
void decoder_plan(Run& r, const mvq_stack& s, const float* z, float* y_out, int B, int t)
{
    const int nb = (int)s.blocks.size();
    const Conv& c0 = s.first;
    float* h = nullptr;
    int rows = t, tv = t, first_block = 0;
    if (decoder_use_packed(s, B, t)) {
        // PACKED latent-rate rows (include/mvq.h): model.0 and the first block's transposed conv on rows of pack_seg segments
        const int per = ceil4(t + 3), seg = s.pack_seg, G = (B + seg - 1) / seg, L = seg * per;
        float* zp = r.ar.alloc((size_t)G * c0.cin * L);
        if (!r.dry) {
            if (hipMemsetAsync(zp, 0, (size_t)G * c0.cin * L * 4, reinterpret_cast<hipStream_t>(r.stream)) != hipSuccess) r.call(sfail(MVQ_EHIP, "decoder_fwd: memset failed"));
            for (int j = 0; j < seg && j < B; ++j)               // items j, j + seg, ... sit at column j * per of consecutive rows
                r.call(mvq_copy3d_f32(z + (size_t)j * c0.cin * t, (size_t)seg * c0.cin * t, (size_t)t, zp + (size_t)j * per, (size_t)c0.cin * L, (size_t)L,
                                      (B - j + seg - 1) / seg, c0.cin, t, r.stream));
        }
        const Block& b0 = s.blocks[0];
        float* hp = r.ar.alloc((size_t)G * c0.cout * L);
        if (!r.dry)
            r.call(mvq_conv1d_packed_rows_f32(zp, c0.wp, c0.bias, nullptr, b0.alpha, hp, nullptr, nullptr, G, c0.cin, c0.cout, 7, 1, 3, MVQ_ACT_NONE, seg, per, t,
                                              r.stream));
        r.ar.release(zp);
        const Conv& up_ = b0.res;
        const int tout = (t - 1) * up_.stride - 2 * up_.pad + 2 * up_.stride;
        const bool wide = b0.width >= s.presnaked_min_c;
        float* u = r.ar.alloc((size_t)B * up_.cout * tout);
        float* us = wide ? r.ar.alloc((size_t)B * up_.cout * tout) : nullptr;
        if (!r.dry)
            r.call(mvq_conv_transpose1d_packed_rows_f32(hp, up_.wp, up_.bias, nullptr, nullptr, u, us, us ? b0.ru[0].alpha_a : nullptr, G, up_.cin, up_.cout,
                                                        up_.stride, up_.pad, seg, per, t, B, r.stream));
        r.ar.release(hp);
        const float* nxt = nb > 1 ? s.blocks[1].alpha : s.alpha_last;
        h = run_units(r, s, b0, B, tout, u, us, nxt, 0);
        rows = tv = tout;
        first_block = 1;
    } else {
        // rows that are not a multiple of 4 (75 latent frames) are carried zero-padded to the next multiple of 4
        const float* zin = z;
        float* zp = nullptr;
        int tvalid = 0;
        rows = t;
        if (t % 4 && t > 0 && c0.cin % 32 == 0) {
            rows = ceil4(t);
            zp = r.ar.alloc((size_t)B * c0.cin * rows);
            if (!r.dry) {
                if (hipMemsetAsync(zp, 0, (size_t)B * c0.cin * rows * 4, reinterpret_cast<hipStream_t>(r.stream)) != hipSuccess) r.call(sfail(MVQ_EHIP, "decoder_fwd: memset failed"));
                r.call(mvq_copy3d_f32(z, (size_t)c0.cin * t, (size_t)t, zp, (size_t)c0.cin * rows, (size_t)rows, B, c0.cin, t, r.stream));
            }
            zin = zp; tvalid = t;
        }
        h = r.ar.alloc((size_t)B * c0.cout * rows);
        if (!r.dry)
            r.call(mvq_conv1d_padded_f32(zin, c0.wp, c0.bias, nullptr, nullptr, s.blocks[0].alpha, h, nullptr, nullptr, B, c0.cin, rows, c0.cout, 7, 1, 1, 3,
                                         MVQ_ACT_NONE, tvalid, r.stream));
        r.ar.release(zp);
        tv = t;
    }
    for (int i = first_block; i < nb && r.ok(); ++i) {
        const float* nxt = i + 1 < nb ? s.blocks[i + 1].alpha : s.alpha_last;
        int tn = 0, rn = 0;
        h = decoder_block(r, s, s.blocks[i], B, h, rows, tv, nxt, &tn, &rn);
        tv = tn; rows = rn;
    }
    if (!r.ok()) return;
    const Conv& cl = s.last;
    if (rows == tv) {
        if (!r.dry)
            r.call(mvq_conv1d_padded_f32(h, cl.wp, cl.bias, nullptr, nullptr, nullptr, y_out, nullptr, nullptr, B, cl.cin, rows, cl.cout, 7, 1, 1, 3, MVQ_ACT_TANH, 0,
                                         r.stream));
    } else {
        float* y = r.ar.alloc((size_t)B * cl.cout * rows);
        if (!r.dry) {
            r.call(mvq_conv1d_padded_f32(h, cl.wp, cl.bias, nullptr, nullptr, nullptr, y, nullptr, nullptr, B, cl.cin, rows, cl.cout, 7, 1, 1, 3, MVQ_ACT_TANH, 0,
                                         r.stream));
            r.call(mvq_copy3d_f32(y, (size_t)cl.cout * rows, (size_t)rows, y_out, (size_t)cl.cout * tv, (size_t)tv, B, cl.cout, tv, r.stream));
        }
        r.ar.release(y);
    }
    r.ar.release(h);
}

// ---- training config: saving forward + backward w.r.t. the input (weights frozen) -------------------------------------------------
// Saved forward, in this order (a bump allocation over `saved`): h0 [model.0 output], per block: u [transposed conv output], then
// per unit: t7 [7-tap pre-activation] and the unit's output; y.  Every tensor is written by its producer kernel directly.
struct SavedLayout {
    struct Blk { size_t x_in, r_x[3], t7[3]; int cin, c, t_in, t; };
    std::vector<Blk> blk;
    size_t hl = 0, y = 0, total = 0;
    int t_out = 0;
};
SavedLayout saved_layout(const mvq_stack& s, int B, int t)
{
    SavedLayout L;
    size_t off = 0;
    auto take = [&](size_t floats) { const size_t o = off; off += up(floats * 4); return o; };
    size_t cur = take((size_t)B * s.first.cout * t);                      // h0
    int tt = t;
    for (auto& b : s.blocks) {
        SavedLayout::Blk k;
        k.cin = b.res.cin; k.c = b.width; k.t_in = tt; k.t = convtr_len(b.res, tt);
        k.x_in = cur;
        const size_t n = (size_t)B * k.c * k.t;
        cur = take(n);                                                    // transposed conv output = input of unit 0
        for (int j = 0; j < 3; ++j) { k.r_x[j] = cur; k.t7[j] = take(n); cur = take(n); }
        L.blk.push_back(k);
        tt = k.t;
    }
    L.hl = cur;
    L.t_out = tt;
    L.y = take((size_t)B * s.last.cout * tt);
    L.total = off;
    return L;
}

void decoder_saving_plan(Run& r, const mvq_stack& s, const float* z, float* y_out, char* saved, int B, int t)
{
    const SavedLayout L = saved_layout(s, B, t);
    auto S = [&](size_t off) { return reinterpret_cast<float*>(saved + off); };
    const int nb = (int)s.blocks.size();
    // every producer emits (raw -> saved, Snake for its consumer -> workspace): no conv evaluates Snake while staging
    float* hs = r.ar.alloc((size_t)B * s.first.cout * t);
    if (!r.dry)
        r.call(mvq_conv1d_padded_f32(z, s.first.wp, s.first.bias, nullptr, nullptr, nullptr, S(L.blk[0].x_in), hs, s.blocks[0].alpha, B, s.first.cin, t, s.first.cout,
                                     7, 1, 1, 3, MVQ_ACT_NONE, 0, r.stream));
    for (int i = 0; i < nb && r.ok(); ++i) {
        const Block& b = s.blocks[i];
        const SavedLayout::Blk& k = L.blk[i];
        const size_t n = (size_t)B * k.c * k.t;
        float* us = r.ar.alloc(n);
        if (!r.dry)
            r.call(mvq_conv_transpose1d_op_f32(hs, b.res.wp, b.res.bias, nullptr, nullptr, S(k.r_x[0]), us, b.ru[0].alpha_a, B, b.res.cin, k.t_in, b.res.cout,
                                               b.res.stride, b.res.pad, b.res.opad, 0, 0, r.stream));
        r.ar.release(hs);
        hs = us;
        for (int j = 0; j < 3; ++j) {
            const Unit& u = b.ru[j];
            float* t7s = r.ar.alloc(n);
            if (!r.dry)
                r.call(mvq_conv1d_padded_f32(hs, u.c7.wp, u.c7.bias, nullptr, nullptr, nullptr, S(k.t7[j]), t7s, u.alpha_b, B, k.c, k.t, k.c, 7, 1, u.c7.dil,
                                             u.c7.pad, MVQ_ACT_NONE, 0, r.stream));
            r.ar.release(hs);
            const float* nxt = j < 2 ? b.ru[j + 1].alpha_a : (i + 1 < nb ? s.blocks[i + 1].alpha : s.alpha_last);
            float* out = S(j < 2 ? k.r_x[j + 1] : (i + 1 < nb ? L.blk[i + 1].x_in : L.hl));
            float* outs = r.ar.alloc(n);
            if (!r.dry)
                r.call(mvq_conv1d_padded_f32(t7s, u.c1.wp, u.c1.bias, nullptr, S(k.r_x[j]), nullptr, out, outs, nxt, B, k.c, k.t, k.c, 1, 1, 1, 0, MVQ_ACT_NONE, 0,
                                             r.stream));
            r.ar.release(t7s);
            hs = outs;
        }
    }
    if (!r.ok()) return;
    if (!r.dry) {
        r.call(mvq_conv1d_padded_f32(hs, s.last.wp, s.last.bias, nullptr, nullptr, nullptr, S(L.y), nullptr, nullptr, B, s.last.cin, L.t_out, s.last.cout, 7, 1, 1, 3,
                                     MVQ_ACT_TANH, 0, r.stream));
        if (hipMemcpyAsync(y_out, S(L.y), (size_t)B * s.last.cout * L.t_out * 4, hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(r.stream)) != hipSuccess)
            r.call(sfail(MVQ_EHIP, "decoder_fwd_saving: copy of y failed"));
    }
    r.ar.release(hs);
}

void decoder_bwd_plan(Run& r, const mvq_stack& s, const char* saved, const float* gy, float* gz, int B, int t)
{
    const SavedLayout L = saved_layout(s, B, t);
    auto S = [&](size_t off) { return reinterpret_cast<const float*>(saved + off); };
    const int nb = (int)s.blocks.size();
    const Conv& cl = s.last;
    const size_t ny = (size_t)B * cl.cout * L.t_out;
    float* g = r.ar.alloc(ny);
    if (!r.dry) r.call(mvq_mul_dtanh_f32(gy, S(L.y), g, ny, r.stream));
    {
        float* g2 = r.ar.alloc((size_t)B * cl.cin * L.t_out);
        if (!r.dry)
            r.call(mvq_conv1d_dgrad_f32(g, cl.wd, S(L.hl), s.alpha_last, nullptr, g2, B, cl.cin, L.t_out, cl.cout, L.t_out, 7, 1, 1, 3, r.stream));
        r.ar.release(g);
        g = g2;
    }
    for (int i = nb - 1; i >= 0 && r.ok(); --i) {
        const Block& b = s.blocks[i];
        const SavedLayout::Blk& k = L.blk[i];
        const size_t n = (size_t)B * k.c * k.t;
        for (int j = 2; j >= 0; --j) {
            const Unit& u = b.ru[j];
            float* g1 = r.ar.alloc(n);
            if (!r.dry)
                r.call(mvq_conv1d_dgrad_f32(g, u.c1.wd, S(k.t7[j]), u.alpha_b, nullptr, g1, B, k.c, k.t, k.c, k.t, 1, 1, 1, 0, r.stream));
            float* g2 = r.ar.alloc(n);
            if (!r.dry)
                r.call(mvq_conv1d_dgrad_f32(g1, u.c7.wd, S(k.r_x[j]), u.alpha_a, g, g2, B, k.c, k.t, k.c, k.t, 7, 1, u.c7.dil, u.c7.pad, r.stream));
            r.ar.release(g1); r.ar.release(g);
            g = g2;
        }
        float* gx = r.ar.alloc((size_t)B * k.cin * k.t_in);
        if (!r.dry)
            r.call(mvq_conv1d_dgrad_f32(g, b.res.wd, S(k.x_in), b.alpha, nullptr, gx, B, k.cin, k.t_in, b.res.cout, k.t, b.res.ks, b.res.stride, 1, b.res.pad,
                                        r.stream));
        r.ar.release(g);
        g = gx;
    }
    if (!r.ok()) return;
    if (!r.dry)
        r.call(mvq_conv1d_dgrad_f32(g, s.first.wd, nullptr, nullptr, nullptr, gz, B, s.first.cin, t, s.first.cout, t, 7, 1, 1, 3, r.stream));
    r.ar.release(g);
}

template <class F> size_t plan_peak(F f)
{
    Run r;
    r.dry = true;
    f(r);
    return r.ar.peak + ALIGN;
}
template <class F> int plan_run(void* ws, size_t ws_bytes, void* stream, const char* what, F f)
{
    Run r;
    r.dry = false;
    r.stream = stream;
    const uintptr_t a = reinterpret_cast<uintptr_t>(ws);
    const size_t shift = (ALIGN - a % ALIGN) % ALIGN;
    if (ws_bytes < shift) return sfail(MVQ_EINVAL, "%s: workspace too small", what);
    r.ar.base = reinterpret_cast<char*>(ws) + shift;
    r.ar.cap = ws_bytes - shift;
    f(r);
    if (r.ar.overflow) return sfail(MVQ_EINVAL, "%s: workspace too small (see the matching *_workspace_bytes query)", what);
    return r.rc;
}

}  // namespace

extern "C" {

int mvq_encoder_create(mvq_stack** out, const mvq_encoder_desc* desc, const float* const* params, void* weights_blob, size_t blob_bytes, void* stream)
{
    if (!out || !desc) return sfail(MVQ_EINVAL, "encoder_create: null argument");
    mvq_stack* s = new mvq_stack();
    int rc = build_encoder(*s, *desc);
    if (rc == MVQ_OK && (params || weights_blob)) {
        if (!params || !weights_blob) rc = sfail(MVQ_EINVAL, "encoder_create: params and weights_blob go together");
        else rc = prepare_weights(*s, params, weights_blob, blob_bytes, stream);
    }
    if (rc != MVQ_OK) { delete s; if (rc == MVQ_EINVAL && !params) sfail(rc, "encoder_create: bad description"); return rc; }
    *out = s;
    return MVQ_OK;
}
int mvq_decoder_create(mvq_stack** out, const mvq_decoder_desc* desc, const float* const* params, void* weights_blob, size_t blob_bytes, void* stream)
{
    if (!out || !desc) return sfail(MVQ_EINVAL, "decoder_create: null argument");
    mvq_stack* s = new mvq_stack();
    int rc = build_decoder(*s, *desc);
    if (rc == MVQ_OK && (params || weights_blob)) {
        if (!params || !weights_blob) rc = sfail(MVQ_EINVAL, "decoder_create: params and weights_blob go together");
        else rc = prepare_weights(*s, params, weights_blob, blob_bytes, stream);
    }
    if (rc != MVQ_OK) { delete s; if (rc == MVQ_EINVAL && !params) sfail(rc, "decoder_create: bad description"); return rc; }
    *out = s;
    return MVQ_OK;
}
void mvq_stack_destroy(mvq_stack* s) { delete s; }

int mvq_stack_param_count(const mvq_stack* s) { return s ? (int)s->names.size() : 0; }
int mvq_stack_param_info(const mvq_stack* s, int i, char* name, int name_len, int* dims3)
{
    if (!s || i < 0 || i >= (int)s->names.size()) return sfail(MVQ_EINVAL, "stack_param_info: index out of range");
    if (name && name_len > 0) snprintf(name, name_len, "%s", s->names[i].c_str());
    if (dims3) for (int k = 0; k < 3; ++k) dims3[k] = k < (int)s->shapes[i].size() ? s->shapes[i][k] : 1;
    return MVQ_OK;
}
size_t mvq_stack_weights_bytes(const mvq_stack* s) { return s ? weights_bytes(*const_cast<mvq_stack*>(s)) : 0; }
int mvq_stack_set_plan(mvq_stack* s, int vpack_min_batch, int pack_min_batch)
{
    if (!s) return sfail(MVQ_EINVAL, "stack_set_plan: null stack");
    if (vpack_min_batch > 0) s->vpack_min_batch = vpack_min_batch;
    if (pack_min_batch > 0) s->pack_min_batch = pack_min_batch;
    return MVQ_OK;
}

int mvq_encoder_out_len(const mvq_stack* s, int t) { return (s && !s->decoder && t > 0) ? encoder_len(*s, t) : 0; }
static bool enc_wants_vpack(const mvq_stack& s, int B)
{
    const Conv& tail = s.last;
    return B >= s.vpack_min_batch && tail.stride == 1 && tail.dil == 1 && 2 * tail.pad == tail.ks - 1 && tail.cin % 32 == 0 && tail.cout % 128 == 0;
}
size_t mvq_encoder_workspace_bytes(const mvq_stack* s, int batch, int t)
{
    if (!s || s->decoder || batch <= 0 || t <= 0) return 0;
    size_t a = plan_peak([&](Run& r) { encoder_plan(r, *s, nullptr, nullptr, batch, t, false); });
    if (enc_wants_vpack(*s, batch)) { const size_t b = plan_peak([&](Run& r) { encoder_plan(r, *s, nullptr, nullptr, batch, t, true); }); a = a > b ? a : b; }
    return a;
}
int mvq_encoder_fwd_f32(const mvq_stack* s, const float* x, float* z, void* workspace, size_t workspace_bytes, int batch, int t, void* stream)
{
    if (!s || s->decoder || !s->first.wp) return sfail(MVQ_EINVAL, "encoder_fwd: not an encoder stack with weights");
    if (batch < 0 || t < 0) return sfail(MVQ_EINVAL, "encoder_fwd: bad shape");
    if (batch == 0 || encoder_len(*s, t) <= 0) return MVQ_OK;
    if (!x || !z || !workspace) return sfail(MVQ_EINVAL, "encoder_fwd: null tensor");
    if (enc_wants_vpack(*s, batch)) {
        const int rc = plan_run(workspace, workspace_bytes, stream, "encoder_fwd", [&](Run& r) { encoder_plan(r, *s, x, z, batch, t, true); });
        if (rc != MVQ_EUNSUPPORTED) return rc;         // no LDS-DMA form for this shape / MVQ_NO_DMA: the plain plan (same results)
        }
    return plan_run(workspace, workspace_bytes, stream, "encoder_fwd", [&](Run& r) { encoder_plan(r, *s, x, z, batch, t, false); });
}

int mvq_decoder_out_len(const mvq_stack* s, int t) { return (s && s->decoder) ? decoder_len(*s, t) : 0; }
size_t mvq_decoder_workspace_bytes(const mvq_stack* s, int batch, int t)
{
    if (!s || !s->decoder || batch <= 0 || t <= 0) return 0;
    size_t a = plan_peak([&](Run& r) { decoder_plan(r, *s, nullptr, nullptr, batch, t); });
    const size_t b = plan_peak([&](Run& r) { decoder_saving_plan(r, *s, nullptr, nullptr, nullptr, batch, t); });
    const size_t c = plan_peak([&](Run& r) { decoder_bwd_plan(r, *s, nullptr, nullptr, nullptr, batch, t); });
    a = a > b ? a : b;
    return a > c ? a : c;
}
int mvq_decoder_fwd_f32(const mvq_stack* s, const float* z, float* y, void* workspace, size_t workspace_bytes, int batch, int t, void* stream)
{
    if (!s || !s->decoder || !s->first.wp) return sfail(MVQ_EINVAL, "decoder_fwd: not a decoder stack with weights");
    if (batch < 0 || t < 0) return sfail(MVQ_EINVAL, "decoder_fwd: bad shape");
    if (batch == 0 || decoder_len(*s, t) <= 0) return MVQ_OK;
    if (!z || !y || !workspace) return sfail(MVQ_EINVAL, "decoder_fwd: null tensor");
    return plan_run(workspace, workspace_bytes, stream, "decoder_fwd", [&](Run& r) { decoder_plan(r, *s, z, y, batch, t); });
}
size_t mvq_decoder_saved_bytes(const mvq_stack* s, int batch, int t)
{
    if (!s || !s->decoder || batch <= 0 || t <= 0) return 0;
    return saved_layout(*s, batch, t).total + ALIGN;
}
static char* align_saved(void* p) { const uintptr_t a = reinterpret_cast<uintptr_t>(p); return reinterpret_cast<char*>(p) + (ALIGN - a % ALIGN) % ALIGN; }
int mvq_decoder_fwd_saving_f32(const mvq_stack* s, const float* z, float* y, void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes,
                               int batch, int t, void* stream)
{
    if (!s || !s->decoder || !s->first.wp) return sfail(MVQ_EINVAL, "decoder_fwd_saving: not a decoder stack with weights");
    if (batch <= 0 || decoder_len(*s, t) <= 0) return sfail(MVQ_EINVAL, "decoder_fwd_saving: empty batch or sequence");
    if (!z || !y || !saved || !workspace) return sfail(MVQ_EINVAL, "decoder_fwd_saving: null tensor");
    if (saved_bytes < mvq_decoder_saved_bytes(s, batch, t)) return sfail(MVQ_EINVAL, "decoder_fwd_saving: saved buffer too small");
    return plan_run(workspace, workspace_bytes, stream, "decoder_fwd_saving", [&](Run& r) { decoder_saving_plan(r, *s, z, y, align_saved(saved), batch, t); });
}
int mvq_decoder_bwd_input_f32(const mvq_stack* s, const void* saved, size_t saved_bytes, const float* gy, float* gz, void* workspace,
                              size_t workspace_bytes, int batch, int t, void* stream)
{
    if (!s || !s->decoder || !s->first.wd) return sfail(MVQ_EINVAL, "decoder_bwd_input: not a decoder stack with weights");
    if (batch <= 0 || decoder_len(*s, t) <= 0) return sfail(MVQ_EINVAL, "decoder_bwd_input: empty batch or sequence");
    if (!saved || !gy || !gz || !workspace) return sfail(MVQ_EINVAL, "decoder_bwd_input: null tensor");
    if (saved_bytes < mvq_decoder_saved_bytes(s, batch, t)) return sfail(MVQ_EINVAL, "decoder_bwd_input: saved buffer too small");
    return plan_run(workspace, workspace_bytes, stream, "decoder_bwd_input",
                    [&](Run& r) { decoder_bwd_plan(r, *s, align_saved(const_cast<void*>(saved)), gy, gz, batch, t); });
}

}  // extern "C"
