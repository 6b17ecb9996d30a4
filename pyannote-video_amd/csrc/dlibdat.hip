// dlibdat.hip -- host-side readers for dlib `.dat` model files (dlib::serialize streams), so that the files the reference passes
// by path (README.md:29-30; face.py:58,62; scripts/pyannote-face.py:37,451-452) load into the same tensors as the `.pvfm`
// container:  shape_predictor_68_face_landmarks.dat -> sp.*      dlib_face_recognition_resnet_model_v1.dat -> emb.*
// [EXT] dlib's source and the real files are not available in this environment; the layout restates serialize.h,
// shape_predictor.h and dnn/{core,layers,tensor}.h as published and is exercised against the Python writer
// (pyannote_video_amd/models.py, tests/test_dlib_dat.py).
//   integer   1 control byte (low nibble = n payload bytes, 0x80 = negative) + n little-endian magnitude bytes
//   float     two integers, mantissa and exponent: value = mantissa * 2^exponent (32000/32001/32002 = +inf/-inf/nan)
//   matrix    integers -nr, -nc, then the elements row-major;  vector / string: integer size, then items / bytes
//   tensor    int version (2), integers n, k, nr, nc, then raw little-endian IEEE floats
#include "pvf_internal.h"
#include <cmath>
#include <fstream>
#include <limits>

namespace {

struct DlibStream {
    const std::vector<uint8_t>& b;
    size_t o = 0;
    std::string path;
    DlibStream(const std::vector<uint8_t>& buf, const std::string& p) : b(buf), path(p) {}
    [[noreturn]] void fail(const char* what) const
    {
        throw PvfError(path + ": not a dlib stream of the expected layout (" + what + " at byte " + std::to_string(o) + ")");
    }
    int64_t integer()
    {
        if (o >= b.size()) fail("unexpected end");
        const uint8_t c = b[o];
        const int n = c & 0x0F;
        if (n == 0 || n > 8 || (c & 0x70) || o + 1 + n > b.size()) fail("bad integer control byte");
        uint64_t v = 0;
        for (int i = n - 1; i >= 0; --i) v = (v << 8) | b[o + 1 + i];
        o += 1 + (size_t)n;
        return (c & 0x80) ? -(int64_t)v : (int64_t)v;
    }
    double real()
    {
        const int64_t m = integer(), e = integer();
        if (e == 32000) return std::numeric_limits<double>::infinity();
        if (e == 32001) return -std::numeric_limits<double>::infinity();
        if (e == 32002) return std::numeric_limits<double>::quiet_NaN();
        return std::ldexp((double)m, (int)e);
    }
    void matrix_f32(std::vector<float>& out, int64_t* nr, int64_t* nc)
    {
        *nr = -integer(); *nc = -integer();
        // every element takes at least two bytes in the stream (mantissa and exponent integers of one byte each plus their control
        // bytes are four; two is a safe floor): a header that promises more elements than the remaining bytes can hold is corrupt, and
        // is refused BEFORE anything is allocated for it
        const uint64_t left = (uint64_t)(b.size() - o);
        if (*nr < 0 || *nc < 0 || (uint64_t)(*nr) > left || (uint64_t)(*nc) > left || (uint64_t)(*nr) * (uint64_t)(*nc) > left / 2) fail("matrix header");
        const size_t n = (size_t)(*nr) * (size_t)(*nc);
        const size_t base = out.size();
        out.resize(base + n);
        for (size_t i = 0; i < n; ++i) out[base + i] = (float)real();
    }
    void tensor(std::vector<float>& out, int64_t dims[4])
    {
        if (integer() != 2) fail("tensor version");
        uint64_t n = 1;
        const uint64_t left = (uint64_t)(b.size() - o);
        for (int i = 0; i < 4; ++i) {
            dims[i] = integer();
            if (dims[i] < 0 || (uint64_t)dims[i] > left) fail("tensor dims");
            n *= (uint64_t)dims[i];                                  // each factor <= left < 2^63 / 4 only after the check below: test as we go
            if (n > left / 4 + 1) fail("tensor dims");
        }
        if (n > (uint64_t)(b.size() - o) / 4) fail("tensor data");
        out.resize(n);
        if (n) memcpy(out.data(), b.data() + o, 4 * n);     // little-endian host
        o += 4 * n;
    }
};

Tensor make_tensor(const std::string& name, int dtype, std::vector<int64_t> dims, const void* data, size_t bytes)
{
    Tensor t;
    t.name = name; t.dtype = dtype; t.dims = std::move(dims);
    t.data.assign((const uint8_t*)data, (const uint8_t*)data + bytes);
    return t;
}

std::vector<uint8_t> slurp(const char* path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw PvfError(std::string("cannot open model file: ") + path);
    const std::streamoff n = f.tellg();
    f.seekg(0);
    std::vector<uint8_t> buf((size_t)n);
    f.read((char*)buf.data(), n);
    if (!f) throw PvfError(std::string("cannot read model file: ") + path);
    return buf;
}

// dlib::shape_predictor: int version(1); matrix<float,0,1> initial_shape; vector<vector<regression_tree>> forests with
// regression_tree = { vector<split_feature{idx1, idx2, thresh}> splits; vector<matrix<float,0,1>> leaf_values };
// vector<vector<unsigned long>> anchor_idx; vector<vector<dlib::vector<float,2>>> deltas
std::map<std::string, Tensor> read_shape_predictor(const char* path)
{
    const std::vector<uint8_t> buf = slurp(path);
    DlibStream s(buf, path);
    if (s.integer() != 1) s.fail("shape_predictor version");
    std::vector<float> initial;
    int64_t nr, nc;
    s.matrix_f32(initial, &nr, &nc);
    const int64_t n_casc = s.integer();
    if (n_casc <= 0 || n_casc > 1000) s.fail("cascade count");
    std::vector<int32_t> idx1, idx2;
    std::vector<float> thresh, leaves;
    int64_t n_trees = -1, n_split = -1, n_leaf = -1, leaf_len = -1;
    for (int64_t c = 0; c < n_casc; ++c) {
        const int64_t nt = s.integer();
        if (n_trees < 0) n_trees = nt;
        if (nt != n_trees || nt <= 0 || nt > 100000) s.fail("tree count");
        for (int64_t t = 0; t < nt; ++t) {
            const int64_t ns = s.integer();
            if (n_split < 0) n_split = ns;
            if (ns != n_split || ns <= 0 || ns > 63) s.fail("split count");
            for (int64_t k = 0; k < ns; ++k) {
                const int64_t i1 = s.integer(), i2 = s.integer();
                if (i1 < 0 || i2 < 0 || i1 > 0x7fffffff || i2 > 0x7fffffff) s.fail("split feature index");     // (< n_pix is checked once n_pix is known)
                idx1.push_back((int32_t)i1);
                idx2.push_back((int32_t)i2);
                thresh.push_back((float)s.real());
            }
            const int64_t nl = s.integer();
            if (n_leaf < 0) n_leaf = nl;
            if (nl != n_leaf || nl != ns + 1) s.fail("leaf count");
            for (int64_t k = 0; k < nl; ++k) {
                int64_t lr, lc;
                s.matrix_f32(leaves, &lr, &lc);
                if (leaf_len < 0) leaf_len = lr * lc;
                if (lr * lc != leaf_len || leaf_len != (int64_t)initial.size()) s.fail("leaf size");
            }
        }
    }
    std::vector<int32_t> anchor;
    int64_t n_pix = -1;
    if (s.integer() != n_casc) s.fail("anchor_idx size");
    for (int64_t c = 0; c < n_casc; ++c) {
        const int64_t n = s.integer();
        if (n_pix < 0) n_pix = n;
        if (n != n_pix || n <= 0 || n > 1024) s.fail("anchor count");
        for (int64_t k = 0; k < n; ++k) {
            const int64_t a = s.integer();
            if (a < 0 || 2 * a >= (int64_t)initial.size()) s.fail("anchor index beyond the shape's parts");
            anchor.push_back((int32_t)a);
        }
    }
    std::vector<float> deltas;
    if (s.integer() != n_casc) s.fail("deltas size");
    for (int64_t c = 0; c < n_casc; ++c) {
        if (s.integer() != n_pix) s.fail("delta count");
        for (int64_t k = 0; k < 2 * n_pix; ++k) deltas.push_back((float)s.real());
    }
    for (size_t k = 0; k < idx1.size(); ++k)
        if (idx1[k] >= n_pix || idx2[k] >= n_pix) s.fail("split feature index beyond the cascade's feature pixels");
    int depth = 0;
    while (((int64_t)1 << depth) < n_leaf) ++depth;
    if (((int64_t)1 << depth) != n_leaf) s.fail("trees are not complete binary trees");
    std::map<std::string, Tensor> m;
    const int32_t meta[5] = {(int32_t)n_casc, (int32_t)n_trees, (int32_t)(initial.size() / 2), (int32_t)n_pix, depth};
    m["sp.meta"] = make_tensor("sp.meta", 1, {5}, meta, sizeof meta);
    m["sp.initial_shape"] = make_tensor("sp.initial_shape", 0, {(int64_t)initial.size()}, initial.data(), initial.size() * 4);
    m["sp.anchor_idx"] = make_tensor("sp.anchor_idx", 1, {n_casc, n_pix}, anchor.data(), anchor.size() * 4);
    m["sp.deltas"] = make_tensor("sp.deltas", 0, {n_casc, n_pix, 2}, deltas.data(), deltas.size() * 4);
    m["sp.split_idx1"] = make_tensor("sp.split_idx1", 1, {n_casc, n_trees, n_split}, idx1.data(), idx1.size() * 4);
    m["sp.split_idx2"] = make_tensor("sp.split_idx2", 1, {n_casc, n_trees, n_split}, idx2.data(), idx2.size() * 4);
    m["sp.split_thresh"] = make_tensor("sp.split_thresh", 0, {n_casc, n_trees, n_split}, thresh.data(), thresh.size() * 4);
    m["sp.leaves"] = make_tensor("sp.leaves", 0, {n_casc, n_trees, n_leaf, leaf_len}, leaves.data(), leaves.size() * 4);
    return m;
}

// [EXT] dlib's mean_face_shape_x / _y (get_face_chip_details): compiled into dlib, not part of the model file
const double MEAN_X[51] = {
    0.000213256, 0.0752622, 0.18113, 0.29077, 0.393397, 0.586856, 0.689483, 0.799124, 0.904991, 0.98004, 0.490127, 0.490127,
    0.490127, 0.490127, 0.36688, 0.426036, 0.490127, 0.554217, 0.613373, 0.121737, 0.187122, 0.265825, 0.334606, 0.260918,
    0.182743, 0.645647, 0.714428, 0.793132, 0.858516, 0.79751, 0.719335, 0.254149, 0.340985, 0.428858, 0.490127, 0.551395,
    0.639268, 0.726104, 0.642159, 0.556721, 0.490127, 0.423532, 0.338094, 0.290379, 0.428096, 0.490127, 0.552157, 0.689874,
    0.553364, 0.490127, 0.42689};
const double MEAN_Y[51] = {
    0.106454, 0.038915, 0.0187482, 0.0344891, 0.0773906, 0.0773906, 0.0344891, 0.0187482, 0.038915, 0.106454, 0.203352,
    0.307009, 0.409805, 0.515625, 0.587326, 0.609345, 0.628106, 0.609345, 0.587326, 0.216423, 0.178758, 0.179852, 0.231733,
    0.245099, 0.244077, 0.231733, 0.179852, 0.178758, 0.216423, 0.244077, 0.245099, 0.780233, 0.745405, 0.727388, 0.742578,
    0.727388, 0.745405, 0.780233, 0.864805, 0.902192, 0.909281, 0.902192, 0.864805, 0.784792, 0.778746, 0.785343, 0.778746,
    0.784792, 0.824182, 0.831803, 0.824182};

// Network stream of anet_type.  The records that wrap the layers (add_layer / tag / skip versions and flags) differ between dlib
// releases; the `details` records of the layers that carry parameters are self-delimiting: a length-prefixed tag string
// ("con_N", "affine_", "fc_N") followed by the `params` tensor, input side first.  They are located by their tags.
std::map<std::string, Tensor> read_embedder(const char* path)
{
    const std::vector<uint8_t> buf = slurp(path);
    struct Rec { size_t after_tag; int kind; };
    std::vector<Rec> recs;
    auto tag_at = [&](size_t o, const char* prefix, int kind) {
        const size_t pl = strlen(prefix);
        if (o < 2 || o + pl > buf.size() || memcmp(buf.data() + o, prefix, pl) != 0 || buf[o - 2] != 0x01) return;
        const size_t ln = buf[o - 1];
        if (ln < pl || ln > pl + 2 || o + ln > buf.size()) return;
        for (size_t k = pl; k < ln; ++k) if (buf[o + k] < '0' || buf[o + k] > '9') return;
        recs.push_back({o + ln, kind});
    };
    for (size_t o = 2; o + 4 <= buf.size(); ++o) {
        if (buf[o] == 'c') tag_at(o, "con_", 0);
        else if (buf[o] == 'a') tag_at(o, "affine_", 1);
        else if (buf[o] == 'f') tag_at(o, "fc_", 2);
        else if (buf[o] == 'b') tag_at(o, "bn_con", 3);          // "bn_con2": a model serialised from the TRAINING net (see below)
    }
    struct Conv { std::vector<float> p; int64_t nf, nr, nc; };
    std::vector<Conv> cons;
    std::vector<std::vector<float>> affs, fcs;
    for (const Rec& r : recs) {
        DlibStream s(buf, path);
        s.o = r.after_tag;
        try {
            int64_t dims[4];
            if (r.kind == 2) { s.integer(); s.integer(); }     // num_outputs, bias_mode
            std::vector<float> p;
            s.tensor(p, dims);
            if (r.kind == 3) {
                // [EXT] bn_ (dlib/dnn/layers.h, "bn_con2"): params (gamma then beta), then the tensors means, invstds, running_means,
                // running_variances, then num_updates, running_stats_window_size, learning-rate / weight-decay multipliers and eps.
                // dlib's affine_ deserialiser accepts such a record and turns it into gamma' = gamma / sqrt(var + eps),
                // beta' = beta - mean * gamma'; the same here, in float like dlib.
                std::vector<float> means, invstds, rmean, rvar;
                s.tensor(means, dims); s.tensor(invstds, dims); s.tensor(rmean, dims); s.tensor(rvar, dims);
                s.integer(); s.integer();                                  // num_updates, running_stats_window_size
                s.real(); s.real(); s.real(); s.real();                    // learning_rate / weight_decay multipliers (+ bias ones)
                const double eps = s.real();
                const size_t ch = p.size() / 2;
                if (p.size() % 2 || rmean.size() != ch || rvar.size() != ch || !(eps > 0 && eps < 1)) throw PvfError("bn record");
                std::vector<float> gb(2 * ch);
                for (size_t k = 0; k < ch; ++k) {
                    const float g = p[k] / std::sqrt(rvar[k] + (float)eps);
                    gb[k] = g;
                    gb[ch + k] = p[ch + k] - rmean[k] * g;
                }
                affs.push_back(std::move(gb));
            } else if (r.kind == 0) {
                Conv c;
                c.nf = s.integer(); c.nr = s.integer(); c.nc = s.integer();
                c.p = std::move(p);
                cons.push_back(std::move(c));
            } else if (r.kind == 1) affs.push_back(std::move(p));
            else fcs.push_back(std::move(p));
        } catch (const PvfError&) { /* a byte pattern that merely looks like a tag */ }
    }
    if (cons.size() != 29 || affs.size() != 29 || fcs.size() != 1)
        throw PvfError(std::string(path) + ": expected 29 con, 29 affine and 1 fc record (dlib anet_type), found " + std::to_string(cons.size()) +
                       " / " + std::to_string(affs.size()) + " / " + std::to_string(fcs.size()));
    static const int UNITS[14][3] = {{32, 32, 0}, {32, 32, 0}, {32, 32, 0}, {32, 64, 1}, {64, 64, 0}, {64, 64, 0}, {64, 64, 0},
                                     {64, 128, 1}, {128, 128, 0}, {128, 128, 0}, {128, 256, 1}, {256, 256, 0}, {256, 256, 0}, {256, 256, 1}};
    std::vector<float> blob;
    auto put = [&](int li, int cin, int cout, int k) {
        const Conv& c = cons[li];
        const size_t nw = (size_t)cout * cin * k * k;
        if (c.nf != cout || c.nr != k || c.nc != k || c.p.size() != nw + (size_t)cout || affs[li].size() != 2 * (size_t)cout)
            throw PvfError(std::string(path) + ": conv layer " + std::to_string(li) + " does not have anet_type's shape");
        blob.insert(blob.end(), c.p.begin(), c.p.end());                 // weights [cout][cin][k][k], then bias
        blob.insert(blob.end(), affs[li].begin(), affs[li].end());       // gamma, beta
    };
    put(0, 3, 32, 7);
    for (int u = 0; u < 14; ++u) { put(1 + 2 * u, UNITS[u][0], UNITS[u][1], 3); put(2 + 2 * u, UNITS[u][1], UNITS[u][1], 3); }
    if (fcs[0].size() != 256 * 128) throw PvfError(std::string(path) + ": fc layer is not 256 x 128");
    blob.insert(blob.end(), fcs[0].begin(), fcs[0].end());
    std::map<std::string, Tensor> m;
    const int32_t meta[1] = {150};
    const double pad[1] = {0.25};
    float mean[102];
    for (int i = 0; i < 51; ++i) { mean[2 * i] = (float)MEAN_X[i]; mean[2 * i + 1] = (float)MEAN_Y[i]; }
    m["emb.meta"] = make_tensor("emb.meta", 1, {1}, meta, sizeof meta);
    m["emb.padding"] = make_tensor("emb.padding", 2, {1}, pad, sizeof pad);
    m["emb.mean_shape"] = make_tensor("emb.mean_shape", 0, {51, 2}, mean, sizeof mean);
    m["emb.blob"] = make_tensor("emb.blob", 0, {(int64_t)blob.size()}, blob.data(), blob.size() * 4);
    return m;
}

} // namespace

// model file given by path: `.pvfm` container or dlib `.dat` stream; kind 1 = shape predictor, 2 = embedder
std::map<std::string, Tensor> pvf_read_model(const char* path, int kind)
{
    char magic[8] = {0};
    {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw PvfError(std::string("cannot open model file: ") + path);
        f.read(magic, 8);
    }
    if (memcmp(magic, "PVFMODEL", 8) == 0) return pvf_read_container(path);
    if (kind == 1) return read_shape_predictor(path);
    if (kind == 2) return read_embedder(path);
    throw PvfError(std::string("unsupported model file: ") + path);
}
