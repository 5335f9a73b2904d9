// beifong_amd host layer — core object model (see core.h).
#include "core.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace bfh {

static LogLevel g_level = Warn;
static int g_gpu_count = 1;
void set_gpu_count(int n) {
    if (n < 1) Throw("gpu count must be >= 1 (got %d)", n);
    g_gpu_count = n;
}
int gpu_count() { return g_gpu_count; }
void set_log_level(LogLevel l) { g_level = l; }

void Throw(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw std::runtime_error(buf);
}
void Log(LogLevel level, const char *fmt, ...) {
    if (level < g_level) return;
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    fprintf(stderr, "[beifong] %s\n", buf);
}

static std::string g_variant = "scalar_rgb";
const std::string &variant() { return g_variant; }
void set_variant(const std::string &v) {
    if (v != "scalar_rgb" && v != "scalar_mono" && v != "scalar_spectral")
        Throw("set_variant(): \"%s\" is not available; built variants: scalar_rgb, scalar_mono, scalar_spectral "
              "(mitsuba.conf:71-77 enables scalar_rgb, scalar_spectral, packet_rgb)", v.c_str());
    g_variant = v;
}

// ---------------------------------------------------------------------------
static Class g_object_class("Object", "", "", nullptr);
const Class *Object::class_() const { return &g_object_class; }
std::string Object::to_string() const { return class_()->name() + "[" + m_id + "]"; }

// ---------------------------------------------------------------------------
Matrix4f Matrix4f::identity() {
    Matrix4f r;
    for (int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.f : 0.f;
    return r;
}
Matrix4f Matrix4f::operator*(const Matrix4f &o) const {
    // enoki matrix product: result column j = sum_k col_k(this) * o(k, j), fma chain from k = 0
    Matrix4f r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = m[4 * i + 0] * o.m[0 + j];
            for (int k = 1; k < 4; ++k) acc = std::fmaf(m[4 * i + k], o.m[4 * k + j], acc);
            r.m[4 * i + j] = acc;
        }
    return r;
}
static bool invert(const Matrix4f &a, Matrix4f &out) {
    double A[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            A[i][j] = a.m[4 * i + j];
            A[i][4 + j] = i == j;
        }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r)
            if (std::fabs(A[r][c]) > std::fabs(A[piv][c])) piv = r;
        if (A[piv][c] == 0.0) return false;
        for (int j = 0; j < 8; ++j) std::swap(A[piv][j], A[c][j]);
        double d = A[c][c];
        for (int j = 0; j < 8; ++j) A[c][j] /= d;
        for (int r = 0; r < 4; ++r)
            if (r != c) {
                double f = A[r][c];
                for (int j = 0; j < 8; ++j) A[r][j] -= f * A[c][j];
            }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out.m[4 * i + j] = (float) A[i][4 + j];
    return true;
}
Transform4f::Transform4f(const Matrix4f &m) : matrix(m) {
    if (!invert(m, inverse)) Throw("Transform4f: singular matrix");
}
Transform4f Transform4f::operator*(const Transform4f &o) const { return Transform4f(matrix * o.matrix, o.inverse * inverse); }
Transform4f Transform4f::translate(Vector3f v) {
    Transform4f t;
    t.matrix.m[3] = v.x; t.matrix.m[7] = v.y; t.matrix.m[11] = v.z;
    t.inverse.m[3] = -v.x; t.inverse.m[7] = -v.y; t.inverse.m[11] = -v.z;
    return t;
}
Transform4f Transform4f::scale(Vector3f v) {
    Transform4f t;
    t.matrix.m[0] = v.x; t.matrix.m[5] = v.y; t.matrix.m[10] = v.z;
    t.inverse.m[0] = 1.f / v.x; t.inverse.m[5] = 1.f / v.y; t.inverse.m[10] = 1.f / v.z;
    return t;
}
Transform4f Transform4f::rotate(Vector3f axis, float angle_deg) {
    // enoki::rotate<Matrix>(axis, deg_to_rad(angle)); the inverse is the transpose
    double n = std::sqrt((double) axis.x * axis.x + (double) axis.y * axis.y + (double) axis.z * axis.z);
    double x = axis.x / n, y = axis.y / n, z = axis.z / n;
    double a = angle_deg * (3.14159265358979323846 / 180.0), s = std::sin(a), c = std::cos(a);
    double r[9] = {c + x * x * (1 - c),     x * y * (1 - c) - z * s, x * z * (1 - c) + y * s,
                   y * x * (1 - c) + z * s, c + y * y * (1 - c),     y * z * (1 - c) - x * s,
                   z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)};
    Transform4f t;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            t.matrix.m[4 * i + j] = (float) r[3 * i + j];
            t.inverse.m[4 * j + i] = (float) r[3 * i + j];
        }
    return t;
}
static Vector3f sub(Vector3f a, Vector3f b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static float dot3(Vector3f a, Vector3f b) { return std::fmaf(a.z, b.z, std::fmaf(a.y, b.y, a.x * b.x)); }
static Vector3f norm3(Vector3f a) {
    float s = 1.f / std::sqrt(dot3(a, a));
    return {a.x * s, a.y * s, a.z * s};
}
static Vector3f cross3(Vector3f a, Vector3f b) {
    return {std::fmaf(a.y, b.z, -(a.z * b.y)), std::fmaf(a.z, b.x, -(a.x * b.z)), std::fmaf(a.x, b.y, -(a.y * b.x))};
}
Transform4f Transform4f::look_at(Vector3f origin, Vector3f target, Vector3f up) {
    // include/mitsuba/core/transform.h:241-268; <lookat> without `up` picks
    // coordinate_system(dir).first (xml.cpp:911-913)
    Vector3f dir = norm3(norm3(sub(target, origin)));
    if (dot3(up, up) == 0.f) {
        float sign = std::copysign(1.f, dir.z), a = -1.f / (sign + dir.z), b = dir.x * dir.y * a;
        up = {sign * (dir.x * dir.x * a) + 1.f, sign * b, -sign * dir.x};
    }
    Vector3f left = norm3(cross3(up, dir));
    Vector3f new_up = cross3(dir, left);
    Transform4f t;
    float *m = t.matrix.m, *iv = t.inverse.m;
    m[0] = left.x; m[4] = left.y; m[8] = left.z;
    m[1] = new_up.x; m[5] = new_up.y; m[9] = new_up.z;
    m[2] = dir.x; m[6] = dir.y; m[10] = dir.z;
    m[3] = origin.x; m[7] = origin.y; m[11] = origin.z;
    iv[0] = left.x; iv[1] = left.y; iv[2] = left.z; iv[3] = -dot3(left, origin);
    iv[4] = new_up.x; iv[5] = new_up.y; iv[6] = new_up.z; iv[7] = -dot3(new_up, origin);
    iv[8] = dir.x; iv[9] = dir.y; iv[10] = dir.z; iv[11] = -dot3(dir, origin);
    for (int i = 0; i < 16; ++i)
        if (std::isnan(m[i])) Throw("invalid lookat transformation");
    return t;
}
Transform4f Transform4f::perspective(float fov, float near_, float far_) {
    // include/mitsuba/core/transform.h:203-220
    float recip = 1.f / (far_ - near_);
    float tn = (float) std::tan((double) (fov * .5f) * (3.14159265358979323846 / 180.0)), cot = 1.f / tn;
    Transform4f t;
    std::memset(t.matrix.m, 0, sizeof(t.matrix.m));
    std::memset(t.inverse.m, 0, sizeof(t.inverse.m));
    t.matrix.m[0] = cot; t.matrix.m[5] = cot; t.matrix.m[10] = far_ * recip;
    t.matrix.m[11] = -near_ * far_ * recip;
    t.matrix.m[14] = 1.f;
    t.inverse.m[0] = tn; t.inverse.m[5] = tn; t.inverse.m[15] = 1.f / near_;
    t.inverse.m[11] = 1.f;
    t.inverse.m[14] = (near_ - far_) / (far_ * near_);
    return t;
}
bool Transform4f::has_scale() const {
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 3; ++j) {
            float sum = 0.f;
            for (int k = 0; k < 3; ++k) sum += matrix.m[4 * i + k] * matrix.m[4 * j + k];
            if (i == j && std::fabs(sum - 1.f) > 1e-3f) return true;
            if (i != j && std::fabs(sum) > 1e-3f) return true;
        }
    return false;
}

// ---------------------------------------------------------------------------
void Properties::put(const std::string &n, Entry e) {
    if (!m_entries.count(n)) m_order.push_back(n);
    m_entries[n] = std::move(e);
}
void Properties::set_bool(const std::string &n, bool v) { Entry e; e.type = Type::Bool; e.b = v; put(n, e); }
void Properties::set_long(const std::string &n, int64_t v) { Entry e; e.type = Type::Long; e.l = v; put(n, e); }
void Properties::set_float(const std::string &n, double v) { Entry e; e.type = Type::Float; e.f = v; put(n, e); }
void Properties::set_string(const std::string &n, const std::string &v) { Entry e; e.type = Type::String; e.s = v; put(n, e); }
void Properties::set_vector3f(const std::string &n, Vector3f v) { Entry e; e.type = Type::Vector; e.v = v; put(n, e); }
void Properties::set_transform(const std::string &n, const Transform4f &v) { Entry e; e.type = Type::Transform; e.t = v; put(n, e); }
void Properties::set_object(const std::string &n, const ref<Object> &v) { Entry e; e.type = Type::Object; e.o = v; put(n, e); }

const Properties::Entry &Properties::get(const std::string &n, Type t) const {
    auto it = m_entries.find(n);
    if (it == m_entries.end()) Throw("Property \"%s\" has not been specified!", n.c_str());
    const Entry &e = it->second;
    bool ok = e.type == t || (t == Type::Float && e.type == Type::Long) || (t == Type::Long && e.type == Type::Float && e.f == (int64_t) e.f);
    if (!ok) Throw("The property \"%s\" has the wrong type.", n.c_str());
    e.queried = true;
    return e;
}
bool Properties::bool_(const std::string &n) const { return get(n, Type::Bool).b; }
bool Properties::bool_(const std::string &n, bool d) const { return has_property(n) ? bool_(n) : d; }
int64_t Properties::int_(const std::string &n) const {
    const Entry &e = get(n, Type::Long);
    return e.type == Type::Long ? e.l : (int64_t) e.f;
}
int64_t Properties::int_(const std::string &n, int64_t d) const { return has_property(n) ? int_(n) : d; }
float Properties::float_(const std::string &n) const {
    const Entry &e = get(n, Type::Float);
    return e.type == Type::Float ? (float) e.f : (float) e.l;
}
float Properties::float_(const std::string &n, float d) const { return has_property(n) ? float_(n) : d; }
std::string Properties::string(const std::string &n) const { return get(n, Type::String).s; }
std::string Properties::string(const std::string &n, const std::string &d) const { return has_property(n) ? string(n) : d; }
Vector3f Properties::vector3f(const std::string &n, Vector3f d) const { return has_property(n) ? get(n, Type::Vector).v : d; }
Transform4f Properties::transform(const std::string &n, const Transform4f &d) const {
    return has_property(n) ? get(n, Type::Transform).t : d;
}
void Properties::mark_queried(const std::string &n) const {
    auto it = m_entries.find(n);
    if (it != m_entries.end()) it->second.queried = true;
}
std::vector<std::pair<std::string, ref<Object>>> Properties::objects(bool mark) const {
    std::vector<std::pair<std::string, ref<Object>>> r;
    for (const auto &n : m_order) {
        const Entry &e = m_entries.at(n);
        if (e.type == Type::Object) {
            if (mark) e.queried = true;
            r.emplace_back(n, e.o);
        }
    }
    return r;
}
std::vector<std::string> Properties::unqueried() const {
    std::vector<std::string> r;
    for (const auto &n : m_order)
        if (!m_entries.at(n).queried) r.push_back(n);
    return r;
}

// ---------------------------------------------------------------------------
static std::map<std::string, const Class *> &registry() {
    static std::map<std::string, const Class *> r;
    return r;
}
static std::set<std::string> &object_tags() {
    static std::set<std::string> t;
    return t;
}
Class::Class(const std::string &name, const std::string &parent, const std::string &variant, ConstructFunctor construct,
             const std::string &alias)
    : m_name(name), m_parent(parent), m_variant(variant), m_alias(alias.empty() ? name : alias), m_construct(construct) {
    registry()[name] = this;
    if (!alias.empty()) object_tags().insert(alias);   // xml.cpp:153-161 register_class
}
ref<Object> Class::construct(const Properties &props) const {
    if (!m_construct) Throw("RTTI error: attempted to construct a class lacking a default constructor (%s)!", m_name.c_str());
    return m_construct(props);
}
bool Class::derives_from(const Class *other) const {
    const Class *c = this;
    while (c) {
        if (c == other) return true;
        if (c->m_parent.empty()) break;
        auto it = registry().find(c->m_parent);
        c = it == registry().end() ? nullptr : it->second;
    }
    return false;
}
const Class *Class::for_name(const std::string &name, const std::string &) {
    auto it = registry().find(name);
    return it == registry().end() ? nullptr : it->second;
}
bool Class::is_object_tag(const std::string &tag) { return object_tags().count(tag) != 0; }

// ---------------------------------------------------------------------------
PluginManager::PluginManager() {
    // plugins/ sits next to libbeifong_host.so (the reference resolves
    // "plugins/<name>.so" through the FileResolver, plugin.cpp:84-116)
    Dl_info info;
    if (dladdr((void *) &set_variant, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t k = p.find_last_of('/');
        m_dir = (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/plugins";
    } else {
        m_dir = "plugins";
    }
}
PluginManager *PluginManager::instance() {
    static PluginManager pm;
    return &pm;
}
void PluginManager::ensure_plugin_loaded(const std::string &name) {
    if (m_plugins.count(name)) return;
    std::string path = m_dir + "/" + name + ".so";
    void *h = dlopen(path.c_str(), RTLD_LAZY | RTLD_LOCAL);     // plugin.cpp:20-41
    if (!h) Throw("Plugin \"%s\" not found! (%s)", name.c_str(), dlerror());
    using StringFunc = const char *(*) ();
    auto pn = (StringFunc) dlsym(h, "plugin_name");
    auto pd = (StringFunc) dlsym(h, "plugin_descr");
    if (!pn || !pd) Throw("Could not resolve symbol \"plugin_name\"/\"plugin_descr\" in \"%s\"", path.c_str());
    // ours, not the reference's: the plugin and this library must have been compiled against the same C-ABI structs
    using AbiFunc = unsigned long long (*)();
    auto pa = (AbiFunc) dlsym(h, "plugin_abi");
    if (!pa || pa() != abi_fingerprint_of_host()) {
        dlclose(h);
        Throw("Plugin \"%s\" was built against another version of include/beifong_hip.h / render.h than libbeifong_host.so "
              "(stale plugins/%s.so): rebuild the host layer (make -C beifong_amd/host)", name.c_str(), name.c_str());
    }
    m_plugins[name] = Plugin{h, pn(), pd()};
    Log(Debug, "Loaded plugin \"%s\" (%s)", pn(), pd());
}
ref<Object> PluginManager::create_object(const Properties &props, const std::string &parent_class) {
    const std::string &name = props.plugin_name();
    if (name == "scene" || name == "ref") {
        const Class *c = Class::for_name(name == "scene" ? "Scene" : name);
        if (!c) Throw("class %s is not registered", name.c_str());
        return c->construct(props);
    }
    ensure_plugin_loaded(name);
    const Class *c = Class::for_name(m_plugins[name].name);
    if (!c) Throw("Plugin \"%s\" did not register a class", name.c_str());
    const Class *parent = parent_class.empty() ? nullptr : Class::for_name(parent_class);
    if (parent && !c->derives_from(parent))                       // plugin.cpp:176-183
        Throw("Type mismatch when loading plugin \"%s\": Expected an instance of type \"%s\", got an instance of type \"%s\"",
              name.c_str(), parent_class.c_str(), c->parent_name().c_str());
    ref<Object> o = c->construct(props);
    o->set_id(props.id());
    return o;
}
std::vector<std::string> PluginManager::loaded_plugins() const {
    std::vector<std::string> r;
    for (auto &p : m_plugins) r.push_back(p.first);
    return r;
}

// ---------------------------------------------------------------------------
// OpenEXR 2.0 single-part scanline writer (NO_COMPRESSION, FLOAT channels)
// ---------------------------------------------------------------------------
namespace {
struct ExrOut {
    std::string buf;
    void bytes(const void *p, size_t n) { buf.append((const char *) p, n); }
    void str(const std::string &s) { buf.append(s.c_str(), s.size() + 1); }
    void i32(int32_t v) { bytes(&v, 4); }
    void u64(uint64_t v) { bytes(&v, 8); }
    void f32(float v) { bytes(&v, 4); }
    void u8(uint8_t v) { bytes(&v, 1); }
    void attr(const char *name, const char *type, const std::string &value) {
        str(name);
        str(type);
        i32((int32_t) value.size());
        buf.append(value);
    }
};
}  // namespace

void write_exr(const std::string &path, uint32_t width, uint32_t height, const std::vector<std::string> &names,
               const float *data) {
    const size_t nc = names.size();
    if (nc == 0 || width == 0 || height == 0) Throw("write_exr(\"%s\"): empty image", path.c_str());
    std::vector<size_t> order(nc);
    for (size_t i = 0; i < nc; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return names[a] < names[b]; });
    for (size_t i = 1; i < nc; ++i)
        if (names[order[i]] == names[order[i - 1]]) Throw("write_exr(\"%s\"): duplicate channel \"%s\"", path.c_str(), names[order[i]].c_str());

    ExrOut o;
    const unsigned char magic[4] = {0x76, 0x2f, 0x31, 0x01};
    o.bytes(magic, 4);
    o.i32(2);                                           // version 2, single-part scanline, short names
    {
        ExrOut ch;
        for (size_t k : order) {
            if (names[k].size() > 31) Throw("write_exr: channel name \"%s\" longer than 31 characters", names[k].c_str());
            ch.str(names[k]);
            ch.i32(2);                                  // FLOAT
            ch.u8(0);                                   // pLinear
            ch.u8(0); ch.u8(0); ch.u8(0);
            ch.i32(1);                                  // x, y sampling
            ch.i32(1);
        }
        ch.u8(0);
        o.attr("channels", "chlist", ch.buf);
    }
    o.attr("compression", "compression", std::string(1, '\0'));
    {
        ExrOut b;
        b.i32(0); b.i32(0); b.i32((int32_t) width - 1); b.i32((int32_t) height - 1);
        o.attr("dataWindow", "box2i", b.buf);
        o.attr("displayWindow", "box2i", b.buf);
    }
    o.attr("lineOrder", "lineOrder", std::string(1, '\0'));
    {
        ExrOut f;
        f.f32(1.f);
        o.attr("pixelAspectRatio", "float", f.buf);
        o.attr("screenWindowWidth", "float", f.buf);
        ExrOut v;
        v.f32(0.f); v.f32(0.f);
        o.attr("screenWindowCenter", "v2f", v.buf);
    }
    o.u8(0);                                            // end of header
    const uint64_t line_bytes = (uint64_t) width * nc * 4, block = 8 + line_bytes;
    const uint64_t first = o.buf.size() + (uint64_t) height * 8;
    for (uint32_t y = 0; y < height; ++y) o.u64(first + y * block);
    std::vector<float> line((size_t) width * nc);
    for (uint32_t y = 0; y < height; ++y) {
        o.i32((int32_t) y);
        o.i32((int32_t) line_bytes);
        for (size_t c = 0; c < nc; ++c)
            for (uint32_t x = 0; x < width; ++x) line[c * width + x] = data[((size_t) y * width + x) * nc + order[c]];
        o.bytes(line.data(), line_bytes);
    }
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) Throw("write_exr: cannot open \"%s\" for writing", path.c_str());
    size_t w = fwrite(o.buf.data(), 1, o.buf.size(), f);
    fclose(f);
    if (w != o.buf.size()) Throw("write_exr: short write to \"%s\"", path.c_str());
}

}  // namespace bfh
