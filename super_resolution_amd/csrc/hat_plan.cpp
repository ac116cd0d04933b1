// hat_plan.cpp — whole-network entry points for hosts without Python (contract: "Forward plans" in include/hat_mi355x.h).
//
// A plan is the complete launch list of one HAT forward for one input shape — every C-ABI call of this library in order,
// with its descriptors / scalars and with every device pointer expressed as (buffer id, byte offset) — plus the buffers
// themselves: packed weights and constants (with their bytes), workspace (zero-initialised), the input and the output.
// `python -m super_resolution_amd.plan` writes it (it records the calls the Python engine makes for that shape, so whatever
// the engine does — fused or unfused kernels, any model variant — is what the plan replays); hat_plan_load() allocates and
// uploads, hat_plan_forward() patches the pointers and issues the same kernels on ONE stream.  No Python, no torch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/hat_mi355x.h"

namespace {

enum : uint32_t { ARG_INT = 0, ARG_FLOAT = 1, ARG_PTR = 2, ARG_STRUCT = 3, ARG_HOST = 4, ARG_STREAM = 5 };
enum : uint32_t { BUF_CONST = 0, BUF_SCRATCH = 1, BUF_INPUT = 2, BUF_OUTPUT = 3 };
constexpr uint32_t NULL_BUF = 0xFFFFFFFFu;

struct Fix { uint32_t field_off, buf; uint64_t off; };
struct Arg {
    uint32_t tag = 0;
    int64_t i = 0;
    double f = 0;
    uint32_t buf = NULL_BUF;
    uint64_t off = 0;
    std::vector<unsigned char> blob;   // struct image / host array
    std::vector<Fix> fix;
};
struct Call { uint32_t fn; std::vector<Arg> args; };
struct Buf { uint32_t kind; uint64_t nbytes; void* dev = nullptr; };

}  // namespace

struct hat_plan {
    int32_t dims[8];   // B, Cin, H, W, scale, Cout, dtype, reserved
    std::vector<Buf> bufs;
    std::vector<Call> calls;
    int device = -1;   // the device the buffers were allocated on (hat_plan_forward refuses to launch on another one)
};

namespace {

struct Reader {
    FILE* f;
    bool ok = true;
    template <typename T> T get() { T v{}; if (fread(&v, sizeof(T), 1, f) != 1) ok = false; return v; }
    void bytes(void* p, size_t n) { if (n && fread(p, 1, n, f) != n) ok = false; }
};

// function ids: the order of this table is the file format (super_resolution_amd/plan.py FN_IDS mirrors it)
const char* const FN_NAMES[] = {"hat_conv", "hat_linear", "hat_conv3x3_small", "hat_cab_fold", "hat_aggr_cab", "hat_ffn", "hat_ffn2",
                                "hat_hab_tail", "hat_layernorm", "hat_esc_weights", "hat_eca_scale", "hat_dwconv_gate", "hat_sgfn_gate",
                                "hat_ocab_attention", "hat_window_attention", "hat_cab_squeeze", "hat_conv3x3_to_planes", "hat_add_f32", "hat_esc_conv13", "hat_ocab_keybias", "hat_ocab_attention_kb", "hat_ocab_mlp", "hat_ocab_qkv", "hat_hab_tail3", "hat_ocab_attention_log2"};
constexpr uint32_t N_FN = sizeof(FN_NAMES) / sizeof(FN_NAMES[0]);

struct Resolved {   // argument values of one call with the pointers patched
    const hat_plan* p;
    const Call* c;
    const void* x;
    void* y;
    void* stream;
    std::vector<std::vector<unsigned char>> tmp;
    void* ptr(uint32_t buf, uint64_t off) const {
        if (buf == NULL_BUF) return nullptr;
        const Buf& b = p->bufs[buf];
        char* base = b.kind == BUF_INPUT ? (char*)const_cast<void*>(x) : (b.kind == BUF_OUTPUT ? (char*)y : (char*)b.dev);
        return base + off;
    }
    int32_t I(size_t k) const { return (int32_t)c->args[k].i; }
    int64_t L(size_t k) const { return c->args[k].i; }
    float F(size_t k) const { return (float)c->args[k].f; }
    void* P(size_t k) {
        const Arg& a = c->args[k];
        if (a.tag == ARG_PTR) return ptr(a.buf, a.off);
        if (a.tag == ARG_STREAM) return stream;
        if (a.tag == ARG_STRUCT || a.tag == ARG_HOST) {
            tmp.emplace_back(a.blob);
            for (const Fix& fx : a.fix) {
                void* v = ptr(fx.buf, fx.off);
                memcpy(tmp.back().data() + fx.field_off, &v, sizeof(void*));
            }
            return tmp.back().data();
        }
        return nullptr;
    }
};

int dispatch(Resolved& r) {
    const size_t n = r.c->args.size();
    switch (r.c->fn) {
        case 0: return n == 2 ? hat_conv((const HatConvDesc*)r.P(0), r.P(1)) : HAT_EINVAL;
        case 1: return n == 2 ? hat_linear((const HatConvDesc*)r.P(0), r.P(1)) : HAT_EINVAL;
        case 2: return n == 2 ? hat_conv3x3_small((const HatConvDesc*)r.P(0), r.P(1)) : HAT_EINVAL;
        case 3: return n == 2 ? hat_cab_fold((const HatCabFoldDesc*)r.P(0), r.P(1)) : HAT_EINVAL;
        case 4: return n == 2 ? hat_aggr_cab((const HatAggrCabDesc*)r.P(0), r.P(1)) : HAT_EINVAL;
        case 5: return n == 2 ? hat_ffn((const HatFfnDesc*)r.P(0), r.P(1)) : HAT_EINVAL;
        case 6: return n == 2 ? hat_ffn2((const HatFfnDesc*)r.P(0), r.P(1)) : HAT_EINVAL;
        case 7: return n == 2 ? hat_hab_tail((const HatHabTailDesc*)r.P(0), r.P(1)) : HAT_EINVAL;
        case 8:
            return n == 13 ? hat_layernorm((const float*)r.P(0), r.P(1), (const float*)r.P(2), (const float*)r.P(3), (float*)r.P(4), r.I(5), r.L(6),
                                           r.I(7), r.I(8), r.I(9), r.I(10), r.I(11), r.P(12))
                           : HAT_EINVAL;
        case 9:
            return n == 15 ? hat_esc_weights((const float*)r.P(0), r.I(1), r.L(2), (const float*)r.P(3), (const float*)r.P(4), (const float*)r.P(5),
                                             (const float*)r.P(6), (const float*)r.P(7), r.P(8), r.I(9), r.I(10), r.I(11), r.I(12), r.I(13), r.P(14))
                           : HAT_EINVAL;
        case 10:
            return n == 12 ? hat_eca_scale((const float*)r.P(0), r.I(1), r.I(2), r.L(3), (const float*)r.P(4), r.I(5), r.F(6), (float*)r.P(7),
                                           (float*)r.P(8), r.I(9), r.I(10), r.P(11))
                           : HAT_EINVAL;
        case 11:
            return n == 12 ? hat_dwconv_gate(r.P(0), (const float*)r.P(1), (const float*)r.P(2), r.P(3), r.I(4), r.I(5), r.I(6), r.I(7), r.I(8), r.I(9),
                                             r.I(10), r.P(11))
                           : HAT_EINVAL;
        case 12:
            return n == 12 ? hat_sgfn_gate(r.P(0), (const float*)r.P(1), (const float*)r.P(2), r.P(3), r.I(4), r.I(5), r.I(6), r.I(7), r.I(8), r.I(9),
                                           r.I(10), r.P(11))
                           : HAT_EINVAL;
        case 13:
            return n == 16 ? hat_ocab_attention(r.P(0), r.P(1), (const float*)r.P(2), r.P(3), r.I(4), r.I(5), r.I(6), r.I(7), r.I(8), r.I(9), r.I(10),
                                                r.I(11), r.I(12), r.I(13), r.I(14), r.P(15))
                           : HAT_EINVAL;
        case 14:
            return n == 16 ? hat_window_attention(r.P(0), r.P(1), (const float*)r.P(2), r.P(3), r.I(4), r.I(5), r.I(6), r.I(7), r.I(8), r.I(9), r.I(10),
                                                  r.I(11), r.I(12), r.I(13), r.I(14), r.P(15))
                           : HAT_EINVAL;
        case 15:
            return n == 12 ? hat_cab_squeeze(r.P(0), r.P(1), (const float*)r.P(2), r.P(3), (float*)r.P(4), r.I(5), r.I(6), r.I(7), r.I(8), r.I(9), r.I(10),
                                             r.P(11))
                           : HAT_EINVAL;
        case 16:
            return n == 14 ? hat_conv3x3_to_planes(r.P(0), r.P(1), (const float*)r.P(2), (float*)r.P(3), r.I(4), r.I(5), r.I(6), r.I(7), r.I(8), r.I(9),
                                                   r.F(10), (const float*)r.P(11), r.I(12), r.P(13))
                           : HAT_EINVAL;
        case 17: return n == 7 ? hat_add_f32((const float*)r.P(0), (const float*)r.P(1), (float*)r.P(2), r.I(3), r.L(4), r.L(5), r.P(6)) : HAT_EINVAL;
        case 18: return n == 10 ? hat_esc_conv13(r.P(0), r.I(1), r.P(2), r.I(3), r.P(4), r.I(5), r.I(6), r.I(7), r.I(8), r.P(9)) : HAT_EINVAL;
        case 19:
            return n == 15 ? hat_ocab_keybias(r.P(0), r.I(1), r.P(2), r.I(3), (float*)r.P(4), r.I(5), r.I(6), r.I(7), r.I(8), r.I(9), r.I(10), r.I(11), r.I(12),
                                              r.I(13), r.P(14))
                           : HAT_EINVAL;
        case 20:
            return n == 18 ? hat_ocab_attention_kb(r.P(0), r.P(1), (const float*)r.P(2), (const float*)r.P(3), r.P(4), r.I(5), r.I(6), r.I(7), r.I(8), r.I(9),
                                                   r.I(10), r.I(11), r.I(12), r.I(13), r.I(14), r.I(15), r.I(16), r.P(17))
                           : HAT_EINVAL;
        case 21: return n == 2 ? hat_ocab_mlp((const HatMlpDesc*)r.P(0), r.P(1)) : HAT_EINVAL;
        case 22: return n == 2 ? hat_ocab_qkv((const HatMlpDesc*)r.P(0), r.P(1)) : HAT_EINVAL;
        case 23: return n == 2 ? hat_hab_tail3((const HatHabTailDesc*)r.P(0), r.P(1)) : HAT_EINVAL;
        case 24:
            return n == 16 ? hat_ocab_attention_log2(r.P(0), r.P(1), (const float*)r.P(2), r.P(3), r.I(4), r.I(5), r.I(6), r.I(7), r.I(8), r.I(9), r.I(10),
                                                     r.I(11), r.I(12), r.I(13), r.I(14), r.P(15))
                           : HAT_EINVAL;
        default: return HAT_EUNSUPPORTED;
    }
}

}  // namespace

extern "C" void hat_plan_free(hat_plan* p) {
    if (!p) return;
    for (Buf& b : p->bufs)
        if (b.dev) (void)hipFree(b.dev);
    delete p;
}

static int plan_load_impl(const char* path, hat_plan** out);

// The file is untrusted input: every count, size, buffer index and offset is checked against what it indexes, and nothing
// thrown by the containers (a corrupt size -> std::bad_alloc / length_error) may cross the extern "C" boundary.
extern "C" int hat_plan_load(const char* path, hat_plan** out) {
    if (!path || !out) return HAT_EINVAL;
    *out = nullptr;
    try {
        return plan_load_impl(path, out);
    } catch (...) {
        return HAT_EINVAL;
    }
}

namespace {
constexpr uint64_t MAX_BUF_BYTES = 1ull << 38;   // 256 GiB: above one MI355X's HBM
// (buffer, offset) must point INTO the buffer (one-past-the-end is allowed for empty views)
bool ptr_ok(const std::vector<Buf>& bufs, uint32_t buf, uint64_t off) {
    if (buf == NULL_BUF) return true;
    return buf < bufs.size() && off <= bufs[buf].nbytes;
}
struct PlanGuard {   // frees a half-built plan on every exit path, exceptions included
    hat_plan* p;
    ~PlanGuard() { if (p) hat_plan_free(p); }
};
}  // namespace

static int plan_load_impl(const char* path, hat_plan** out) {
    FILE* f = fopen(path, "rb");
    if (!f) return HAT_EINVAL;
    Reader rd{f};
    char magic[8];
    rd.bytes(magic, 8);
    if (!rd.ok || memcmp(magic, "HATPLAN1", 8) != 0) { fclose(f); return HAT_EINVAL; }
    const uint32_t version = rd.get<uint32_t>(), nbuf = rd.get<uint32_t>(), ncall = rd.get<uint32_t>(), nfn = rd.get<uint32_t>();
    // version = HAT_ABI_VERSION of the writer: the calls embed raw images of this header's descriptors
    if (!rd.ok || version != HAT_ABI_VERSION || nfn != N_FN || nbuf > (1u << 20) || ncall > (1u << 24)) { fclose(f); return HAT_EUNSUPPORTED; }
    hat_plan* p = new hat_plan();
    PlanGuard guard{p};
    struct FileGuard { FILE* f; ~FileGuard() { if (f) fclose(f); } } fguard{f};
    if (hipGetDevice(&p->device) != hipSuccess) return HAT_EINVAL;
    for (int i = 0; i < 8; ++i) p->dims[i] = rd.get<int32_t>();
    int rc = 0;
    for (uint32_t i = 0; i < nbuf && rd.ok && !rc; ++i) {
        Buf b;
        b.kind = rd.get<uint32_t>();
        (void)rd.get<uint32_t>();
        b.nbytes = rd.get<uint64_t>();
        if (!rd.ok || b.kind > BUF_OUTPUT || b.nbytes > MAX_BUF_BYTES) { rc = HAT_EINVAL; break; }
        if (b.kind == BUF_CONST || b.kind == BUF_SCRATCH) {
            hipError_t e = hipMalloc(&b.dev, b.nbytes ? b.nbytes : 16);
            if (e != hipSuccess) { rc = (int)e; break; }
            if (b.kind == BUF_CONST) {
                std::vector<unsigned char> host(b.nbytes);
                rd.bytes(host.data(), b.nbytes);
                const uint64_t padn = (8 - b.nbytes % 8) % 8;
                unsigned char padb[8];
                rd.bytes(padb, padn);
                e = hipMemcpy(b.dev, host.data(), b.nbytes, hipMemcpyHostToDevice);
            } else {
                e = hipMemset(b.dev, 0, b.nbytes ? b.nbytes : 16);
            }
            if (e != hipSuccess) rc = (int)e;
        }
        p->bufs.push_back(b);
    }
    for (uint32_t i = 0; i < ncall && rd.ok && !rc; ++i) {
        Call c;
        c.fn = rd.get<uint32_t>();
        const uint32_t nargs = rd.get<uint32_t>();
        if (c.fn >= N_FN || nargs > 32) { rc = HAT_EINVAL; break; }
        for (uint32_t k = 0; k < nargs && rd.ok; ++k) {
            Arg a;
            a.tag = rd.get<uint32_t>();
            (void)rd.get<uint32_t>();
            if (a.tag == ARG_INT) a.i = rd.get<int64_t>();
            else if (a.tag == ARG_FLOAT) a.f = rd.get<double>();
            else if (a.tag == ARG_PTR) { a.buf = rd.get<uint32_t>(); (void)rd.get<uint32_t>(); a.off = rd.get<uint64_t>(); }
            else if (a.tag == ARG_STRUCT || a.tag == ARG_HOST) {
                const uint32_t nb = rd.get<uint32_t>(), nfix = rd.get<uint32_t>();
                if (nb > (1u << 20) || nfix > 64) { rc = HAT_EINVAL; break; }
                a.blob.resize((nb + 7) / 8 * 8);
                rd.bytes(a.blob.data(), a.blob.size());
                for (uint32_t j = 0; j < nfix; ++j) {
                    Fix fx;
                    fx.field_off = rd.get<uint32_t>();
                    fx.buf = rd.get<uint32_t>();
                    fx.off = rd.get<uint64_t>();
                    if ((uint64_t)fx.field_off + sizeof(void*) > nb || !ptr_ok(p->bufs, fx.buf, fx.off)) rc = HAT_EINVAL;
                    a.fix.push_back(fx);
                }
            } else if (a.tag != ARG_STREAM) rc = HAT_EINVAL;
            if (a.tag == ARG_PTR && !ptr_ok(p->bufs, a.buf, a.off)) rc = HAT_EINVAL;
            c.args.push_back(std::move(a));
        }
        p->calls.push_back(std::move(c));
    }
    if (!rd.ok && !rc) rc = HAT_EINVAL;
    if (!rc) rc = (int)hipDeviceSynchronize();
    if (rc) return rc;       // (the guards free the plan and close the file)
    guard.p = nullptr;
    *out = p;
    return 0;
}

extern "C" int hat_plan_info(const hat_plan* p, int32_t* dims8, int64_t* n_calls, int64_t* device_bytes) {
    if (!p) return HAT_EINVAL;
    if (dims8) memcpy(dims8, p->dims, sizeof(p->dims));
    if (n_calls) *n_calls = (int64_t)p->calls.size();
    if (device_bytes) {
        int64_t s = 0;
        for (const Buf& b : p->bufs)
            if (b.dev) s += (int64_t)b.nbytes;
        *device_bytes = s;
    }
    return 0;
}

extern "C" int hat_plan_forward(const hat_plan* p, const float* x, float* y, void* stream) {
    if (!p || !x || !y) return HAT_EINVAL;
    int dev = -1;
    // the kernels launch on the CURRENT device: it must be the one the plan's buffers live on
    if (hipGetDevice(&dev) != hipSuccess || dev != p->device) return HAT_EINVAL;
    for (const Call& c : p->calls) {
        Resolved r{p, &c, x, y, stream, {}};
        const int rc = dispatch(r);
        if (rc) return rc;
    }
    return 0;
}
