// lt_archive.cpp -- the reference's checkpoint payloads, read and written without libtorch.
//
// PPOLearner::SaveTo / LoadFrom (PRIV/PPO/PPOLearner.cpp:372-477) store each network with torch::save(nn::Sequential, stream)
// and each optimizer with optim::Adam::save(OutputArchive).  Both are "TorchScript module" zip archives:
//     <dir>/data.pkl        pickle (protocol 2) of the module object tree; tensors are persistent ids naming a storage record
//     <dir>/data/<key>      raw little-endian storage bytes
//     <dir>/code/...        one class definition per object type (attribute names and types)
//     <dir>/constants.pkl   pickled ()      <dir>/version  "3\n"      <dir>/byteorder  "little"
// Model:      object { "0": Linear{weight,bias}, "1": ReLU{}, "2": Linear, ..., "<2L-2>": Linear }   (DiscretePolicy.cpp:10-26)
// Optimizer:  object { pytorch_version "1.5.0", state { <key>: {step:int, exp_avg, exp_avg_sq} ... },
//                      param_groups { "param_groups/size", "param_groups/0": { "params/size", "params/<i>": <key>, options{lr,betas,eps,weight_decay,amsgrad} } } }
//             (torch/optim/serialize.h; keys are arbitrary strings, params/<i> gives the order)
// Host-only code (g++): part of librlgpu.so so that both hosts (learner.py through ctypes, host/Learner.hip) share it.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rlgpu.h"

namespace {

thread_local std::string g_lt_error;

[[noreturn]] void fail(const std::string& s) { throw std::runtime_error(s); }

// ---------------------------------------------------------------------------------------------------------------- zip
uint32_t crc32_of(const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

template <class T>
void put(std::vector<uint8_t>& o, T v) { for (size_t i = 0; i < sizeof(T); i++) o.push_back((uint8_t)((uint64_t)v >> (8 * i))); }
void put_bytes(std::vector<uint8_t>& o, const void* p, size_t n) { o.insert(o.end(), (const uint8_t*)p, (const uint8_t*)p + n); }

// Stored (uncompressed) entries; tensor records start on 64-byte boundaries like the reference's writer pads them ("FB" extra field)
struct ZipWriter {
    std::vector<uint8_t> out, central;
    int n_entries = 0;
    void add(const std::string& name, const void* data, size_t n) {
        const uint32_t crc = crc32_of((const uint8_t*)data, n);
        const size_t hdr_at = out.size();
        size_t data_at = hdr_at + 30 + name.size() + 4;
        const size_t pad = (64 - data_at % 64) % 64;
        const uint16_t extra_len = (uint16_t)(4 + pad);
        put<uint32_t>(out, 0x04034b50); put<uint16_t>(out, 20); put<uint16_t>(out, 0x0800); put<uint16_t>(out, 0);
        put<uint16_t>(out, 0); put<uint16_t>(out, 0x0021);   // time, date (1980-01-01)
        put<uint32_t>(out, crc); put<uint32_t>(out, (uint32_t)n); put<uint32_t>(out, (uint32_t)n);
        put<uint16_t>(out, (uint16_t)name.size()); put<uint16_t>(out, extra_len);
        put_bytes(out, name.data(), name.size());
        put<uint16_t>(out, 0x4246); put<uint16_t>(out, (uint16_t)pad);
        for (size_t i = 0; i < pad; i++) out.push_back('Z');
        put_bytes(out, data, n);
        put<uint32_t>(central, 0x02014b50); put<uint16_t>(central, 20); put<uint16_t>(central, 20); put<uint16_t>(central, 0x0800); put<uint16_t>(central, 0);
        put<uint16_t>(central, 0); put<uint16_t>(central, 0x0021);
        put<uint32_t>(central, crc); put<uint32_t>(central, (uint32_t)n); put<uint32_t>(central, (uint32_t)n);
        put<uint16_t>(central, (uint16_t)name.size()); put<uint16_t>(central, 0); put<uint16_t>(central, 0);
        put<uint16_t>(central, 0); put<uint16_t>(central, 0); put<uint32_t>(central, 0); put<uint32_t>(central, (uint32_t)hdr_at);
        put_bytes(central, name.data(), name.size());
        n_entries++;
    }
    void add(const std::string& name, const std::string& s) { add(name, s.data(), s.size()); }
    void finish(const char* path) {
        const size_t cd_at = out.size();
        put_bytes(out, central.data(), central.size());
        put<uint32_t>(out, 0x06054b50); put<uint16_t>(out, 0); put<uint16_t>(out, 0); put<uint16_t>(out, (uint16_t)n_entries); put<uint16_t>(out, (uint16_t)n_entries);
        put<uint32_t>(out, (uint32_t)central.size()); put<uint32_t>(out, (uint32_t)cd_at); put<uint16_t>(out, 0);
        FILE* f = fopen(path, "wb");
        if (!f) fail(std::string("cannot open ") + path + " for writing");
        const size_t w = fwrite(out.data(), 1, out.size(), f);
        if (fclose(f) != 0 || w != out.size()) fail(std::string("short write to ") + path);
    }
};

struct ZipReader {
    std::vector<uint8_t> buf;
    struct Entry { uint16_t method; uint64_t csize, usize, local_at; };
    std::map<std::string, Entry> entries;
    std::string prefix;   // "<dir>/"

    template <class T>
    T get(size_t at) const {
        if (at + sizeof(T) > buf.size()) fail("truncated zip archive");
        uint64_t v = 0;
        for (size_t i = 0; i < sizeof(T); i++) v |= (uint64_t)buf[at + i] << (8 * i);
        return (T)v;
    }
    explicit ZipReader(const char* path) {
        FILE* f = fopen(path, "rb");
        if (!f) fail(std::string("file ") + path + " does not exist or can't be accessed");
        fseek(f, 0, SEEK_END);
        long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        buf.resize(n > 0 ? (size_t)n : 0);
        if (n > 0 && fread(buf.data(), 1, (size_t)n, f) != (size_t)n) { fclose(f); fail(std::string("short read from ") + path); }
        fclose(f);
        if (buf.size() < 22) fail(std::string(path) + " is not a zip archive");
        // end of central directory, searched backwards (there may be a trailing comment)
        size_t eocd = std::string::npos;
        for (size_t i = buf.size() - 22;; i--) {
            if (get<uint32_t>(i) == 0x06054b50) { eocd = i; break; }
            if (i == 0 || buf.size() - i > 22 + 65535) break;
        }
        if (eocd == std::string::npos) fail(std::string(path) + " is not a zip archive (no end-of-central-directory record)");
        uint64_t count = get<uint16_t>(eocd + 10), cd_size = get<uint32_t>(eocd + 12), cd_at = get<uint32_t>(eocd + 16);
        if (eocd >= 20 && get<uint32_t>(eocd - 20) == 0x07064b50) {   // zip64 locator -> zip64 end record (the reference's writer always emits one)
            const uint64_t z = get<uint64_t>(eocd - 20 + 8);
            if (get<uint32_t>((size_t)z) != 0x06064b50) fail("bad zip64 end-of-central-directory record");
            count = get<uint64_t>((size_t)z + 32); cd_size = get<uint64_t>((size_t)z + 40); cd_at = get<uint64_t>((size_t)z + 48);
        }
        size_t p = (size_t)cd_at;
        for (uint64_t e = 0; e < count; e++) {
            if (get<uint32_t>(p) != 0x02014b50) fail("bad central directory entry");
            Entry en;
            en.method = get<uint16_t>(p + 10);
            en.csize = get<uint32_t>(p + 20); en.usize = get<uint32_t>(p + 24);
            const uint16_t nl = get<uint16_t>(p + 28), el = get<uint16_t>(p + 30), cl = get<uint16_t>(p + 32);
            en.local_at = get<uint32_t>(p + 42);
            if (p + 46 + nl > buf.size()) fail("truncated zip archive");
            std::string name((const char*)&buf[p + 46], nl);
            size_t x = p + 46 + nl;
            const size_t xe = x + el;
            while (x + 4 <= xe) {   // zip64 extended information: only the fields that overflowed, in this order
                const uint16_t id = get<uint16_t>(x), sz = get<uint16_t>(x + 2);
                if (id == 0x0001) {
                    size_t q = x + 4;
                    if (en.usize == 0xFFFFFFFFu) { en.usize = get<uint64_t>(q); q += 8; }
                    if (en.csize == 0xFFFFFFFFu) { en.csize = get<uint64_t>(q); q += 8; }
                    if (en.local_at == 0xFFFFFFFFu) { en.local_at = get<uint64_t>(q); q += 8; }
                }
                x += 4 + sz;
            }
            entries[name] = en;
            p += 46 + nl + el + cl;
        }
        (void)cd_size;
        for (auto& kv : entries) {
            const size_t k = kv.first.find("/data.pkl");
            if (k != std::string::npos && k + 9 == kv.first.size() && kv.first.find('/') == k) { prefix = kv.first.substr(0, k + 1); break; }
        }
        if (prefix.empty()) fail(std::string(path) + " has no <dir>/data.pkl record: not a TorchScript archive");
    }
    // pointer to a stored record
    const uint8_t* record(const std::string& name, size_t& n) const {
        auto it = entries.find(prefix + name);
        if (it == entries.end()) fail("archive has no record " + prefix + name);
        const Entry& e = it->second;
        if (e.method != 0) fail("record " + name + " is compressed; tensor and pickle records are expected stored");
        const size_t l = (size_t)e.local_at;
        if (get<uint32_t>(l) != 0x04034b50) fail("bad local file header");
        const size_t at = l + 30 + get<uint16_t>(l + 26) + get<uint16_t>(l + 28);
        if (at + e.usize > buf.size()) fail("truncated zip archive");
        n = (size_t)e.usize;
        return &buf[at];
    }
};

// ------------------------------------------------------------------------------------------------------------- pickle
struct PVal;
using P = std::shared_ptr<PVal>;
struct PVal {
    enum Kind { NONE, BOOL, INT, FLOAT, STR, TUPLE, LIST, DICT, GLOBAL, OBJECT, PERSID, REDUCE, MARK } kind = NONE;
    int64_t i = 0; double f = 0; std::string s;    // GLOBAL: "module name"
    std::vector<P> items;                           // TUPLE/LIST; REDUCE: {callable, args}; PERSID: {id}; OBJECT: {class, args, state}
    std::vector<std::pair<P, P>> dict;              // DICT, insertion-ordered
    const PVal* find(const std::string& key) const {
        for (auto& kv : dict) if (kv.first->kind == STR && kv.first->s == key) return kv.second.get();
        return nullptr;
    }
};
P mk(PVal::Kind k) { auto p = std::make_shared<PVal>(); p->kind = k; return p; }

// the subset of protocol 2 that torch's pickler emits (torch/csrc/jit/serialization/pickler.cpp)
P unpickle(const uint8_t* d, size_t n) {
    std::vector<P> st;
    std::map<uint32_t, P> memo;
    size_t p = 0;
    auto need = [&](size_t k) { if (p + k > n) fail("truncated pickle"); };
    auto rd = [&](int bytes) { need(bytes); uint64_t v = 0; for (int i = 0; i < bytes; i++) v |= (uint64_t)d[p + i] << (8 * i); p += bytes; return v; };
    auto pop = [&]() { if (st.empty()) fail("pickle stack underflow"); P v = st.back(); st.pop_back(); return v; };
    auto pop_mark = [&]() {
        std::vector<P> v;
        for (;;) { P x = pop(); if (x->kind == PVal::MARK) break; v.push_back(x); }
        std::reverse(v.begin(), v.end());
        return v;
    };
    auto line = [&]() { std::string s; for (;;) { need(1); char c = (char)d[p++]; if (c == '\n') break; s += c; } return s; };
    for (;;) {
        need(1);
        const uint8_t op = d[p++];
        switch (op) {
        case 0x80: rd(1); break;                                            // PROTO
        case '.': return pop();                                             // STOP
        case '(': st.push_back(mk(PVal::MARK)); break;
        case 'N': st.push_back(mk(PVal::NONE)); break;
        case 0x88: case 0x89: { P v = mk(PVal::BOOL); v->i = op == 0x88; st.push_back(v); break; }
        case 'K': { P v = mk(PVal::INT); v->i = (int64_t)rd(1); st.push_back(v); break; }
        case 'M': { P v = mk(PVal::INT); v->i = (int64_t)rd(2); st.push_back(v); break; }
        case 'J': { P v = mk(PVal::INT); v->i = (int32_t)rd(4); st.push_back(v); break; }
        case 0x8a: {                                                        // LONG1
            const int k = (int)rd(1);
            if (k > 8) fail("pickle: integer wider than 64 bits");
            uint64_t u = rd(k);
            if (k > 0 && k < 8 && (u >> (8 * k - 1)) & 1) u |= ~0ull << (8 * k);
            P v = mk(PVal::INT); v->i = (int64_t)u; st.push_back(v); break;
        }
        case 'G': { need(8); uint64_t u = 0; for (int i = 0; i < 8; i++) u = (u << 8) | d[p + i]; p += 8; P v = mk(PVal::FLOAT); memcpy(&v->f, &u, 8); st.push_back(v); break; }
        case 'X': { const size_t k = (size_t)rd(4); need(k); P v = mk(PVal::STR); v->s.assign((const char*)d + p, k); p += k; st.push_back(v); break; }
        case 0x8c: { const size_t k = (size_t)rd(1); need(k); P v = mk(PVal::STR); v->s.assign((const char*)d + p, k); p += k; st.push_back(v); break; }
        case 'c': { P v = mk(PVal::GLOBAL); v->s = line(); v->s += ' '; v->s += line(); st.push_back(v); break; }
        case 'q': memo[(uint32_t)rd(1)] = st.empty() ? (fail("pickle stack underflow"), P()) : st.back(); break;
        case 'r': memo[(uint32_t)rd(4)] = st.empty() ? (fail("pickle stack underflow"), P()) : st.back(); break;
        case 'h': case 'j': { const uint32_t k = (uint32_t)rd(op == 'h' ? 1 : 4); auto it = memo.find(k); if (it == memo.end()) fail("pickle: unknown memo id"); st.push_back(it->second); break; }
        case ')': st.push_back(mk(PVal::TUPLE)); break;
        case ']': st.push_back(mk(PVal::LIST)); break;
        case '}': st.push_back(mk(PVal::DICT)); break;
        case 't': { P v = mk(PVal::TUPLE); v->items = pop_mark(); st.push_back(v); break; }
        case 0x85: case 0x86: case 0x87: {
            P v = mk(PVal::TUPLE); v->items.resize(op - 0x84);
            for (int i = op - 0x85; i >= 0; i--) v->items[i] = pop();
            st.push_back(v); break;
        }
        case 'a': { P x = pop(); if (st.empty() || st.back()->kind != PVal::LIST) fail("pickle: APPEND to a non-list"); st.back()->items.push_back(x); break; }
        case 'e': { auto v = pop_mark(); if (st.empty() || st.back()->kind != PVal::LIST) fail("pickle: APPENDS to a non-list"); for (auto& x : v) st.back()->items.push_back(x); break; }
        case 's': { P val = pop(), key = pop(); if (st.empty() || st.back()->kind != PVal::DICT) fail("pickle: SETITEM on a non-dict"); st.back()->dict.push_back({key, val}); break; }
        case 'u': {
            auto v = pop_mark();
            if (st.empty() || st.back()->kind != PVal::DICT || v.size() % 2) fail("pickle: SETITEMS on a non-dict");
            for (size_t i = 0; i < v.size(); i += 2) st.back()->dict.push_back({v[i], v[i + 1]});
            break;
        }
        case 'Q': { P v = mk(PVal::PERSID); v->items.push_back(pop()); st.push_back(v); break; }
        case 'R': {
            P args = pop(), fn = pop();
            if (fn->kind == PVal::GLOBAL && fn->s == "collections OrderedDict") { st.push_back(mk(PVal::DICT)); break; }
            P v = mk(PVal::REDUCE); v->items = {fn, args}; st.push_back(v); break;
        }
        case 0x81: { P args = pop(), cls = pop(); P v = mk(PVal::OBJECT); v->items = {cls, args, mk(PVal::NONE)}; st.push_back(v); break; }
        case 'b': { P state = pop(); if (st.empty() || st.back()->kind != PVal::OBJECT) fail("pickle: BUILD on a non-object"); st.back()->items[2] = state; break; }
        default: { char b[64]; snprintf(b, sizeof b, "pickle: unsupported opcode 0x%02x at %zu", op, p - 1); fail(b); }
        }
    }
}

const PVal& obj_state(const PVal* v, const char* what) {
    if (!v || v->kind != PVal::OBJECT || v->items[2]->kind != PVal::DICT) fail(std::string("archive: ") + what + " is not a module object");
    return *v->items[2];
}

struct TensorRef { std::string key, dtype; int64_t offset = 0, numel_storage = 0; std::vector<int64_t> sizes, strides; };

TensorRef as_tensor(const PVal* v, const char* what) {
    auto bad = [&]() { fail(std::string("archive: ") + what + " is not a tensor"); };
    if (!v || v->kind != PVal::REDUCE || v->items[0]->kind != PVal::GLOBAL || v->items[0]->s != "torch._utils _rebuild_tensor_v2") bad();
    const PVal& a = *v->items[1];
    if (a.kind != PVal::TUPLE || a.items.size() < 4 || a.items[0]->kind != PVal::PERSID) bad();
    const PVal& pid = *a.items[0]->items[0];
    if (pid.kind != PVal::TUPLE || pid.items.size() < 5 || pid.items[1]->kind != PVal::GLOBAL || pid.items[2]->kind != PVal::STR) bad();
    TensorRef t;
    t.dtype = pid.items[1]->s; t.key = pid.items[2]->s; t.numel_storage = pid.items[4]->i;
    t.offset = a.items[1]->i;
    for (auto& x : a.items[2]->items) t.sizes.push_back(x->i);
    for (auto& x : a.items[3]->items) t.strides.push_back(x->i);
    return t;
}

// copy a (<= 2-D) float tensor out of the archive, honouring its strides
void read_float_tensor(const ZipReader& z, const TensorRef& t, const std::vector<int64_t>& want, float* out, const std::string& what) {
    if (t.dtype != "torch FloatStorage") fail("archive: " + what + " is stored as " + t.dtype + ", expected float32");
    if (t.sizes != want) {
        std::string a, b;
        for (auto v : t.sizes) a += std::to_string(v) + " ";
        for (auto v : want) b += std::to_string(v) + " ";
        fail("Saved model has different size than current model: " + what + " saved [ " + a + "], current [ " + b + "]");
    }
    size_t n = 0;
    const uint8_t* d = z.record("data/" + t.key, n);
    const int64_t rows = want.size() == 2 ? want[0] : 1, cols = want.empty() ? 1 : want.back();
    const int64_t rs = want.size() == 2 ? t.strides[0] : 0, cs = want.empty() ? 0 : t.strides.back();
    for (int64_t r = 0; r < rows; r++)
        for (int64_t c = 0; c < cols; c++) {
            const int64_t e = t.offset + r * rs + c * cs;
            if (e < 0 || (size_t)(e + 1) * 4 > n) fail("archive: " + what + " points outside its storage record");
            memcpy(&out[r * cols + c], d + (size_t)e * 4, 4);
        }
}

// ------------------------------------------------------------------------------------------------------ pickle writing
struct Pickler {
    std::string o;
    void op(char c) { o += c; }
    void str(const std::string& s) { op('X'); uint32_t n = (uint32_t)s.size(); o.append((const char*)&n, 4); o += s; }
    void integer(int64_t v) {
        if (v >= 0 && v < 256) { op('K'); o += (char)v; }
        else if (v >= 0 && v < 65536) { op('M'); uint16_t u = (uint16_t)v; o.append((const char*)&u, 2); }
        else if (v >= INT32_MIN && v <= INT32_MAX) { op('J'); int32_t u = (int32_t)v; o.append((const char*)&u, 4); }
        else { op((char)0x8a); o += (char)8; o.append((const char*)&v, 8); }
    }
    void real(double v) { op('G'); uint64_t u; memcpy(&u, &v, 8); for (int i = 7; i >= 0; i--) o += (char)(u >> (8 * i)); }
    void boolean(bool b) { op(b ? (char)0x88 : (char)0x89); }
    void global(const std::string& mod, const std::string& name) { op('c'); o += mod; o += '\n'; o += name; o += '\n'; }
    void begin_object(const std::string& cls_module) { global(cls_module, "Module"); op(')'); op((char)0x81); op('}'); op('('); }
    void end_object() { op('u'); op('b'); }
    void tensor(const char* storage, const std::string& key, int64_t numel, const std::vector<int64_t>& sizes, bool requires_grad) {
        global("torch._utils", "_rebuild_tensor_v2");
        op('('); op('(');
        str("storage"); global("torch", storage); str(key); str("cpu"); integer(numel);
        op('t'); op('Q');
        integer(0);
        op('('); for (auto s : sizes) integer(s); op('t');
        op('('); for (size_t i = 0; i < sizes.size(); i++) { int64_t st = 1; for (size_t k = i + 1; k < sizes.size(); k++) st *= sizes[k]; integer(st); } op('t');
        boolean(requires_grad);
        global("collections", "OrderedDict"); op(')'); op('R');
        op('t'); op('R');
    }
};

std::string mangle(int i) { return "__torch__.___torch_mangle_" + std::to_string(i); }
std::string mangle_file(int i) { return "code/__torch__/___torch_mangle_" + std::to_string(i) + ".py"; }
const char CLASS_HEAD[] = "class Module(Module):\n";

void add_common_records(ZipWriter& z, const std::string& dir) {
    z.add(dir + "constants.pkl", std::string("\x80\x02).", 4));
    z.add(dir + "version", std::string("3\n"));
    z.add(dir + "byteorder", std::string("little"));
}

void check_dims(const int32_t* dims, int n_linear) {
    if (!dims || n_linear < 1 || n_linear > 64) fail("bad layer description");
    for (int i = 0; i <= n_linear; i++) if (dims[i] <= 0) fail("bad layer description");
}

void write_model(const char* path, const int32_t* dims, int n_linear, const float* params) {
    check_dims(dims, n_linear);
    const std::string dir = "archive/";
    ZipWriter z;
    Pickler pk;
    pk.op((char)0x80); pk.o += (char)2;
    pk.begin_object("__torch__");
    std::string top = std::string(CLASS_HEAD) + "  __parameters__ = []\n  __buffers__ = []\n  __annotations__ = []\n";
    const float* p = params;
    const int n_modules = 2 * n_linear - 1;
    std::vector<std::pair<std::string, std::string>> code;
    int rec = 0;
    for (int m = 0; m < n_modules; m++) {
        top += "  __annotations__[\"" + std::to_string(m) + "\"] = " + mangle(m) + ".Module\n";
        pk.str(std::to_string(m));
        pk.begin_object(mangle(m));
        if (m % 2 == 0) {   // Linear: weight [out][in], bias [out]
            const int l = m / 2, in = dims[l], out = dims[l + 1];
            pk.str("weight"); pk.tensor("FloatStorage", std::to_string(rec), (int64_t)in * out, {out, in}, true);
            z.add(dir + "data/" + std::to_string(rec++), p, (size_t)in * out * 4); p += (size_t)in * out;
            pk.str("bias"); pk.tensor("FloatStorage", std::to_string(rec), out, {out}, true);
            z.add(dir + "data/" + std::to_string(rec++), p, (size_t)out * 4); p += out;
            code.push_back({mangle_file(m), std::string(CLASS_HEAD) + "  __parameters__ = [\"weight\", \"bias\", ]\n  __buffers__ = []\n  weight : Tensor\n  bias : Tensor\n"});
        } else {            // ReLU
            code.push_back({mangle_file(m), std::string(CLASS_HEAD) + "  __parameters__ = []\n  __buffers__ = []\n"});
        }
        pk.end_object();
    }
    pk.end_object();
    pk.op('.');
    z.add(dir + "data.pkl", pk.o);
    z.add(dir + "code/__torch__.py", top);
    for (auto& c : code) z.add(dir + c.first, c.second);
    add_common_records(z, dir);
    z.finish(path);
}

void read_model(const char* path, const int32_t* dims, int n_linear, float* out) {
    check_dims(dims, n_linear);
    ZipReader z(path);
    size_t n = 0;
    const uint8_t* d = z.record("data.pkl", n);
    P root = unpickle(d, n);
    const PVal& top = obj_state(root.get(), "the root");
    // the parameterised children, in registration order (Sequential: "0", "1", ...)
    std::vector<const PVal*> linears;
    for (auto& kv : top.dict)
        if (kv.second->kind == PVal::OBJECT && kv.second->items[2]->kind == PVal::DICT && kv.second->items[2]->find("weight")) linears.push_back(kv.second->items[2].get());
    if ((int)linears.size() != n_linear)
        fail("Saved model has different size than current model: " + std::to_string(linears.size()) + " linear layers saved, " + std::to_string(n_linear) + " current");
    float* p = out;
    for (int l = 0; l < n_linear; l++) {
        const std::string nm = std::to_string(2 * l);
        read_float_tensor(z, as_tensor(linears[l]->find("weight"), "weight"), {dims[l + 1], dims[l]}, p, nm + ".weight"); p += (size_t)dims[l] * dims[l + 1];
        read_float_tensor(z, as_tensor(linears[l]->find("bias"), "bias"), {dims[l + 1]}, p, nm + ".bias"); p += dims[l + 1];
    }
}

void write_adam(const char* path, const int32_t* dims, int n_linear, float lr, const float* m, const float* v, int64_t step) {
    check_dims(dims, n_linear);
    std::string dir = path;   // OutputArchive::save_to names the directory after the file (PPOLearner.cpp:468-472)
    size_t s = dir.find_last_of("/\\");
    if (s != std::string::npos) dir = dir.substr(s + 1);
    s = dir.find_last_of('.');
    if (s != std::string::npos && s > 0) dir = dir.substr(0, s);
    dir += "/";
    const int n_params = 2 * n_linear;
    ZipWriter z;
    Pickler pk;
    std::vector<std::pair<std::string, std::string>> code;
    auto key_of = [](int i) { return std::to_string(94000000000000LL + 1024LL * i); };   // stand-ins for the parameter addresses libtorch uses as keys
    pk.op((char)0x80); pk.o += (char)2;
    pk.begin_object("__torch__");
    pk.str("pytorch_version"); pk.str("1.5.0");
    int cls = 0, rec = 0;
    // state: one {step, exp_avg, exp_avg_sq} per parameter; a never-stepped optimizer has no state at all (Adam creates it lazily)
    pk.str("state"); pk.begin_object(mangle(cls));
    std::string state_code = std::string(CLASS_HEAD) + "  __parameters__ = []\n  __buffers__ = []\n";
    const int state_cls = cls++;
    if (step > 0) {
        state_code += "  __annotations__ = []\n";
        size_t off = 0;
        for (int i = 0; i < n_params; i++) {
            const int l = i / 2;
            std::vector<int64_t> sizes = (i % 2 == 0) ? std::vector<int64_t>{dims[l + 1], dims[l]} : std::vector<int64_t>{dims[l + 1]};
            const int64_t numel = (i % 2 == 0) ? (int64_t)dims[l] * dims[l + 1] : dims[l + 1];
            state_code += "  __annotations__[\"" + key_of(i) + "\"] = " + mangle(cls) + ".Module\n";
            pk.str(key_of(i)); pk.begin_object(mangle(cls));
            pk.str("step"); pk.integer(step);
            pk.str("exp_avg"); pk.tensor("FloatStorage", std::to_string(rec), numel, sizes, false);
            z.add(dir + "data/" + std::to_string(rec++), m + off, (size_t)numel * 4);
            pk.str("exp_avg_sq"); pk.tensor("FloatStorage", std::to_string(rec), numel, sizes, false);
            z.add(dir + "data/" + std::to_string(rec++), v + off, (size_t)numel * 4);
            pk.end_object();
            code.push_back({mangle_file(cls), std::string(CLASS_HEAD) + "  __parameters__ = []\n  __buffers__ = []\n  step : int\n  exp_avg : Tensor\n  exp_avg_sq : Tensor\n"});
            cls++;
            off += (size_t)numel;
        }
    }
    pk.end_object();
    code.push_back({mangle_file(state_cls), state_code});
    // param_groups: one group listing the parameter keys in order, plus the Adam options
    const int groups_cls = cls++, group_cls = cls++, opt_cls = cls++;
    const int64_t one = 1, np64 = n_params;
    pk.str("param_groups"); pk.begin_object(mangle(groups_cls));
    pk.str("param_groups/size"); pk.tensor("LongStorage", std::to_string(rec), 1, {}, false);
    z.add(dir + "data/" + std::to_string(rec++), &one, 8);
    pk.str("param_groups/0"); pk.begin_object(mangle(group_cls));
    pk.str("params/size"); pk.tensor("LongStorage", std::to_string(rec), 1, {}, false);
    z.add(dir + "data/" + std::to_string(rec++), &np64, 8);
    std::string group_code = std::string(CLASS_HEAD) + "  __parameters__ = [\"params/size\", ]\n  __buffers__ = []\n  __annotations__ = []\n  __annotations__[\"params/size\"] = Tensor\n";
    for (int i = 0; i < n_params; i++) {
        pk.str("params/" + std::to_string(i)); pk.str(key_of(i));
        group_code += "  __annotations__[\"params/" + std::to_string(i) + "\"] = str\n";
    }
    group_code += "  options : " + mangle(opt_cls) + ".Module\n";
    pk.str("options"); pk.begin_object(mangle(opt_cls));
    pk.str("lr"); pk.real((double)lr);
    pk.str("betas"); pk.op('('); pk.real(0.9); pk.real(0.999); pk.op('t');
    pk.str("eps"); pk.real(1e-8);
    pk.str("weight_decay"); pk.real(0.0);
    pk.str("amsgrad"); pk.boolean(false);
    pk.end_object();
    pk.end_object();
    pk.end_object();
    pk.end_object();
    pk.op('.');
    code.push_back({mangle_file(groups_cls), std::string(CLASS_HEAD) + "  __parameters__ = [\"param_groups/size\", ]\n  __buffers__ = []\n  __annotations__ = []\n  __annotations__[\"param_groups/size\"] = Tensor\n  __annotations__[\"param_groups/0\"] = " + mangle(group_cls) + ".Module\n"});
    code.push_back({mangle_file(group_cls), group_code});
    code.push_back({mangle_file(opt_cls), std::string(CLASS_HEAD) + "  __parameters__ = []\n  __buffers__ = []\n  lr : float\n  betas : Tuple[float, float]\n  eps : float\n  weight_decay : float\n  amsgrad : bool\n"});
    z.add(dir + "data.pkl", pk.o);
    z.add(dir + "code/__torch__.py", std::string(CLASS_HEAD) + "  __parameters__ = []\n  __buffers__ = []\n  pytorch_version : str\n  state : " + mangle(state_cls) + ".Module\n  param_groups : " + mangle(groups_cls) + ".Module\n");
    for (auto& c : code) z.add(dir + c.first, c.second);
    add_common_records(z, dir);
    z.finish(path);
}

void read_adam(const char* path, const int32_t* dims, int n_linear, float* m, float* v, int64_t* step) {
    check_dims(dims, n_linear);
    ZipReader z(path);
    size_t n = 0;
    const uint8_t* d = z.record("data.pkl", n);
    P root = unpickle(d, n);
    const PVal& top = obj_state(root.get(), "the root");
    const PVal& state = obj_state(top.find("state"), "state");
    const PVal& groups = obj_state(top.find("param_groups"), "param_groups");
    const PVal& group = obj_state(groups.find("param_groups/0"), "param_groups/0");
    const int n_params = 2 * n_linear;
    int saved = 0;
    while (group.find("params/" + std::to_string(saved))) saved++;
    if (saved != n_params) fail("saved optimizer has " + std::to_string(saved) + " parameters, the current one " + std::to_string(n_params));
    size_t off = 0;
    *step = 0;
    for (int i = 0; i < n_params; i++) {
        const int l = i / 2;
        std::vector<int64_t> sizes = (i % 2 == 0) ? std::vector<int64_t>{dims[l + 1], dims[l]} : std::vector<int64_t>{dims[l + 1]};
        const size_t numel = (i % 2 == 0) ? (size_t)dims[l] * dims[l + 1] : (size_t)dims[l + 1];
        const PVal* key = group.find("params/" + std::to_string(i));
        const PVal* st = key->kind == PVal::STR ? state.find(key->s) : nullptr;
        if (!st) {   // a parameter that was never stepped has no state (optimizer.h: state is created on first use)
            std::fill(m + off, m + off + numel, 0.f); std::fill(v + off, v + off + numel, 0.f);
        } else {
            const PVal& ps = obj_state(st, "a parameter state");
            const std::string nm = "optimizer state of parameter " + std::to_string(i);
            read_float_tensor(z, as_tensor(ps.find("exp_avg"), "exp_avg"), sizes, m + off, nm);
            read_float_tensor(z, as_tensor(ps.find("exp_avg_sq"), "exp_avg_sq"), sizes, v + off, nm);
            const PVal* sp = ps.find("step");
            if (sp && sp->kind == PVal::INT) *step = std::max<int64_t>(*step, sp->i);
        }
        off += numel;
    }
}

template <class F>
int guarded(F&& f) {
    try { f(); g_lt_error.clear(); return RLGPU_OK; }
    catch (const std::exception& e) { g_lt_error = e.what(); return RLGPU_ERR_ARG; }
}

}  // namespace

extern "C" {
int rlgpu_lt_write_model(const char* path, const int32_t* dims, int n_linear, const float* params) {
    return guarded([&] { if (!path || !params) fail("null argument"); write_model(path, dims, n_linear, params); });
}
int rlgpu_lt_read_model(const char* path, const int32_t* dims, int n_linear, float* params_out) {
    return guarded([&] { if (!path || !params_out) fail("null argument"); read_model(path, dims, n_linear, params_out); });
}
int rlgpu_lt_write_adam(const char* path, const int32_t* dims, int n_linear, float lr, const float* exp_avg, const float* exp_avg_sq, int64_t step) {
    return guarded([&] { if (!path || !exp_avg || !exp_avg_sq) fail("null argument"); write_adam(path, dims, n_linear, lr, exp_avg, exp_avg_sq, step); });
}
int rlgpu_lt_read_adam(const char* path, const int32_t* dims, int n_linear, float* exp_avg, float* exp_avg_sq, int64_t* step) {
    return guarded([&] { if (!path || !exp_avg || !exp_avg_sq || !step) fail("null argument"); read_adam(path, dims, n_linear, exp_avg, exp_avg_sq, step); });
}
const char* rlgpu_lt_last_error(void) { return g_lt_error.c_str(); }
}
