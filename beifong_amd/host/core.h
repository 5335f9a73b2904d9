// beifong_amd host layer — core object model.
//
// Mirrors the SOURCE-LEVEL surface of Mitsuba 2's libcore that the radar hot
// path needs (the reference's binary plugin ABI is C++ templates over enoki
// types and cannot be reproduced without enoki):
//
//   Object / ref<T>        include/mitsuba/core/object.h
//   Class registry         include/mitsuba/core/class.h:195-211, src/libcore/class.cpp
//   Properties             include/mitsuba/core/properties.h (typed values, queried-key tracking)
//   PluginManager          src/libcore/plugin.cpp:20-41,84-116,163-185
//                          (dlopen plugins/<type>.so, RTLD_LAZY|RTLD_LOCAL,
//                          resolve plugin_name / plugin_descr, Class::for_name)
//   Transform4f            include/mitsuba/core/transform.h (matrix + inverse)
//   Throw / Log            include/mitsuba/core/logger.h
//
// Everything is plain float (the scalar variants); the compute behind
// Integrator::render / receive is the HIP library (include/beifong_hip.h).
#pragma once
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace bfh {

[[noreturn]] void Throw(const char *fmt, ...);
enum LogLevel { Trace = 0, Debug = 100, Info = 200, Warn = 300, Error = 400 };
void Log(LogLevel level, const char *fmt, ...);
void set_log_level(LogLevel level);
/// GPUs a render / receive call is sharded over (ours: the reference is single-host TBB; bfrender --gpus N, Python
/// set_gpu_count): sample shards per GPU, one RCCL all-reduce of the histogram (bf_render_sharded).  Default 1.
void set_gpu_count(int n);
int gpu_count();

// ---------------------------------------------------------------------------
// Object + intrusive reference counting
// ---------------------------------------------------------------------------
class Class;
class Object {
public:
    Object() = default;
    Object(const Object &) {}
    void inc_ref() const { ++m_ref_count; }
    void dec_ref(bool dealloc = true) const {
        if (--m_ref_count == 0 && dealloc) delete this;
    }
    int ref_count() const { return m_ref_count; }
    virtual const Class *class_() const;
    virtual std::string id() const { return m_id; }
    void set_id(const std::string &id) { m_id = id; }
    virtual std::string to_string() const;

protected:
    virtual ~Object() = default;

private:
    mutable int m_ref_count = 0;
    std::string m_id;
};

template <typename T> class ref {
public:
    ref() = default;
    ref(T *p) : m_ptr(p) {
        if (m_ptr) ((Object *) m_ptr)->inc_ref();
    }
    ref(const ref &r) : m_ptr(r.m_ptr) {
        if (m_ptr) ((Object *) m_ptr)->inc_ref();
    }
    ref(ref &&r) noexcept : m_ptr(r.m_ptr) { r.m_ptr = nullptr; }
    ~ref() {
        if (m_ptr) ((Object *) m_ptr)->dec_ref();
    }
    ref &operator=(const ref &r) {
        if (r.m_ptr) ((Object *) r.m_ptr)->inc_ref();
        if (m_ptr) ((Object *) m_ptr)->dec_ref();
        m_ptr = r.m_ptr;
        return *this;
    }
    ref &operator=(T *p) { return *this = ref(p); }
    T *operator->() const { return m_ptr; }
    T &operator*() const { return *m_ptr; }
    T *get() const { return m_ptr; }
    operator T *() const { return m_ptr; }
    explicit operator bool() const { return m_ptr != nullptr; }

private:
    T *m_ptr = nullptr;
};

// ---------------------------------------------------------------------------
// Transform4f — matrix and inverse are composed side by side, exactly as
// include/mitsuba/core/transform.h does (so to_object is not a numeric inverse)
// ---------------------------------------------------------------------------
struct Vector3f {
    float x = 0, y = 0, z = 0;
};
struct Matrix4f {
    float m[16];    // row-major
    static Matrix4f identity();
    Matrix4f operator*(const Matrix4f &o) const;
};
struct Transform4f {
    Matrix4f matrix = Matrix4f::identity(), inverse = Matrix4f::identity();
    Transform4f() = default;
    explicit Transform4f(const Matrix4f &m);               // numeric inverse (transform.h ctor)
    Transform4f(const Matrix4f &m, const Matrix4f &inv) : matrix(m), inverse(inv) {}
    Transform4f operator*(const Transform4f &o) const;     // transform.h operator*
    static Transform4f translate(Vector3f v);
    static Transform4f scale(Vector3f v);
    static Transform4f rotate(Vector3f axis, float angle_deg);
    static Transform4f look_at(Vector3f origin, Vector3f target, Vector3f up);
    static Transform4f perspective(float fov_deg, float near_, float far_);
    bool has_scale() const;
};

// ---------------------------------------------------------------------------
// Properties
// ---------------------------------------------------------------------------
class Properties {
public:
    enum class Type { Bool, Long, Float, String, Vector, Transform, Object };
    struct Entry {
        Type type;
        bool b = false;
        int64_t l = 0;
        double f = 0;
        std::string s;
        Vector3f v;
        Transform4f t;
        ref<Object> o;
        mutable bool queried = false;
    };
    Properties() = default;
    explicit Properties(const std::string &plugin_name) : m_plugin_name(plugin_name) {}
    const std::string &plugin_name() const { return m_plugin_name; }
    void set_plugin_name(const std::string &n) { m_plugin_name = n; }
    const std::string &id() const { return m_id; }
    void set_id(const std::string &id) { m_id = id; }

    bool has_property(const std::string &name) const { return m_entries.count(name) != 0; }
    void set_bool(const std::string &n, bool v);
    void set_long(const std::string &n, int64_t v);
    void set_float(const std::string &n, double v);
    void set_string(const std::string &n, const std::string &v);
    void set_vector3f(const std::string &n, Vector3f v);
    void set_transform(const std::string &n, const Transform4f &v);
    void set_object(const std::string &n, const ref<Object> &v);

    bool bool_(const std::string &n) const;
    bool bool_(const std::string &n, bool def) const;
    int64_t int_(const std::string &n) const;
    int64_t int_(const std::string &n, int64_t def) const;
    float float_(const std::string &n) const;
    float float_(const std::string &n, float def) const;
    std::string string(const std::string &n) const;
    std::string string(const std::string &n, const std::string &def) const;
    Vector3f vector3f(const std::string &n, Vector3f def) const;
    Transform4f transform(const std::string &n, const Transform4f &def = Transform4f()) const;
    /// constant-spectrum "texture" lookup: a float, or a uniform spectrum object
    float texture_value(const std::string &n, float def) const;
    bool has_texture(const std::string &n) const { return has_property(n); }

    /// all Object-typed entries, in insertion order (props.objects())
    std::vector<std::pair<std::string, ref<Object>>> objects(bool mark_queried = true) const;
    void mark_queried(const std::string &n) const;
    std::vector<std::string> unqueried() const;

private:
    const Entry &get(const std::string &n, Type t) const;
    std::string m_plugin_name, m_id;
    std::map<std::string, Entry> m_entries;
    std::vector<std::string> m_order;
    void put(const std::string &n, Entry e);
};

// ---------------------------------------------------------------------------
// Class registry + PluginManager
// ---------------------------------------------------------------------------
class Class {
public:
    using ConstructFunctor = Object *(*) (const Properties &);
    Class(const std::string &name, const std::string &parent, const std::string &variant, ConstructFunctor construct,
          const std::string &alias = "");
    const std::string &name() const { return m_name; }
    const std::string &parent_name() const { return m_parent; }
    const std::string &variant() const { return m_variant; }
    const std::string &alias() const { return m_alias; }
    bool is_constructible() const { return m_construct != nullptr; }
    ref<Object> construct(const Properties &props) const;
    bool derives_from(const Class *other) const;
    static const Class *for_name(const std::string &name, const std::string &variant = "");
    /// XML tag alias ("bsdf", "shape", "transmitter", ...) -> registered? (xml.cpp:153-161)
    static bool is_object_tag(const std::string &tag);

private:
    std::string m_name, m_parent, m_variant, m_alias;
    ConstructFunctor m_construct;
};

class PluginManager {
public:
    static PluginManager *instance();
    /// plugin.cpp:163-185 — load plugins/<type>.so if needed, construct, check the base class
    ref<Object> create_object(const Properties &props, const std::string &parent_class);
    void ensure_plugin_loaded(const std::string &name);
    std::vector<std::string> loaded_plugins() const;
    void set_plugin_dir(const std::string &dir) { m_dir = dir; }
    const std::string &plugin_dir() const { return m_dir; }

private:
    PluginManager();
    struct Plugin {
        void *handle;
        std::string name, descr;
    };
    std::map<std::string, Plugin> m_plugins;
    std::string m_dir;
};

/// current variant ("scalar_rgb", "scalar_mono", "scalar_spectral")
const std::string &variant();
void set_variant(const std::string &v);

/// C-ABI layout word of the headers THIS translation unit was compiled against (include/beifong_hip.h:
/// BF_ABI_FINGERPRINT) mixed with the size of the host classes that embed C-ABI structs.  The PluginManager refuses a
/// plugin whose word differs from the host library's: a stale plugins/<x>.so would construct objects with another
/// layout than libbeifong_host.so reads (the segfault of round 2, DESIGN.md "ABI handshake").
unsigned long long abi_fingerprint_of_host();            // as libbeifong_host.so was compiled
inline unsigned long long abi_fingerprint_of_this_build();

// every plugin translation unit ends with this (class.h:195-211: the reference's
// MTS_EXPORT_PLUGIN emits the same two extern "C" symbols; plugin_abi is ours)
#define BF_EXPORT_PLUGIN(ClassName, ParentName, PluginName, Descr)                                            \
    extern "C" {                                                                                              \
    __attribute__((visibility("default"))) const char *plugin_name() { return PluginName; }                  \
    __attribute__((visibility("default"))) const char *plugin_descr() { return Descr; }                      \
    __attribute__((visibility("default"))) unsigned long long plugin_abi() { return ::bfh::abi_fingerprint_of_this_build(); } \
    }                                                                                                         \
    static ::bfh::Object *bf_construct_##ClassName(const ::bfh::Properties &p) { return new ClassName(p); }  \
    static ::bfh::Class bf_class_##ClassName(PluginName, ParentName, "", bf_construct_##ClassName);

/// Multi-channel float32 OpenEXR 2 scanline file, uncompressed — what Bitmap::write(OpenEXR) produces for the
/// raw film / ADC storage (hdrfilm.cpp:213-249, hdradc.cpp:276-295), minus the compression.  `data` is
/// [height][width][channels] interleaved; channels are stored under `names` (EXR keeps them in alphabetical order).
void write_exr(const std::string &path, uint32_t width, uint32_t height, const std::vector<std::string> &names,
               const float *data);

}  // namespace bfh
