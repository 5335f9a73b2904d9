// beifong_amd host layer — render classes (see render.h).
#include "render.h"

#include <chrono>
#include <cstring>

namespace bfh {

// ---- class registry (class.h:195-211; aliases are the XML tags, xml.cpp:153-161)
static Class c_texture("Texture", "Object", "", nullptr, "texture");
static Class c_rfilter("ReconstructionFilter", "Object", "", nullptr, "rfilter");
static Class c_sampler("Sampler", "Object", "", nullptr, "sampler");
static Class c_bsdf("BSDF", "Object", "", nullptr, "bsdf");
static Class c_film("Film", "Object", "", nullptr, "film");
static Class c_adc("ADC", "Object", "", nullptr, "adc");
static Class c_endpoint("Endpoint", "Object", "", nullptr);
static Class c_emitter("Emitter", "Endpoint", "", nullptr, "emitter");
static Class c_transmitter("Transmitter", "Endpoint", "", nullptr, "transmitter");
static Class c_sensor("Sensor", "Endpoint", "", nullptr, "sensor");
static Class c_receiver("Receiver", "Endpoint", "", nullptr, "receiver");
static Class c_shape("Shape", "Object", "", nullptr, "shape");
static Class c_mesh("Mesh", "Shape", "", nullptr);
static Class c_integrator("Integrator", "Object", "", nullptr, "integrator");
static Class c_sampling_integrator("SamplingIntegrator", "Integrator", "", nullptr);
static Object *construct_scene(const Properties &p) { return new Scene(p); }
static Class c_scene("Scene", "Object", "", construct_scene, "scene");

const Class *Texture::class_() const { return &c_texture; }
const Class *ReconstructionFilter::class_() const { return &c_rfilter; }
const Class *Sampler::class_() const { return &c_sampler; }
const Class *BSDF::class_() const { return &c_bsdf; }
const Class *Film::class_() const { return &c_film; }
const Class *ADC::class_() const { return &c_adc; }
const Class *Endpoint::class_() const { return &c_endpoint; }
const Class *Emitter::class_() const { return &c_emitter; }
const Class *Transmitter::class_() const { return &c_transmitter; }
const Class *Sensor::class_() const { return &c_sensor; }
const Class *Receiver::class_() const { return &c_receiver; }
const Class *Shape::class_() const { return &c_shape; }
const Class *Mesh::class_() const { return &c_mesh; }
const Class *Integrator::class_() const { return &c_integrator; }
const Class *SamplingIntegrator::class_() const { return &c_sampling_integrator; }
const Class *Scene::class_() const { return &c_scene; }

float Properties::texture_value(const std::string &n, float def) const {
    auto it = m_entries.find(n);
    if (it == m_entries.end()) return def;
    const Entry &e = it->second;
    e.queried = true;
    if (e.type == Type::Float) return (float) e.f;
    if (e.type == Type::Long) return (float) e.l;
    if (e.type == Type::Object) {
        auto *t = dynamic_cast<Texture *>(e.o.get());
        if (t) return t->value();
    }
    Throw("The property \"%s\" must be a float or a (uniform) spectrum.", n.c_str());
}

// ---- small objects ----------------------------------------------------------
Sampler::Sampler(const Properties &props) {
    m_sample_count = (size_t) props.int_("sample_count", 4);    // sampler.cpp:11-16
    m_base_seed = (uint64_t) props.int_("seed", 0);
}

static constexpr float kRayEpsilon = 1500 * 5.9604644775390625e-8f;      // math::RayEpsilon<float>
void ReconstructionFilter::init_discretization() {                        // rfilter.cpp:9-21
    if (!(m_radius > 0.f)) Throw("ReconstructionFilter: radius must be positive");
    for (size_t i = 0; i < BF_FILTER_RESOLUTION; ++i) m_values[i] = eval((m_radius * i) / BF_FILTER_RESOLUTION);
    m_values[BF_FILTER_RESOLUTION] = 0;
    m_scale_factor = BF_FILTER_RESOLUTION / m_radius;
    m_border_size = (uint32_t) (int) std::ceil(m_radius - .5f - 2.f * kRayEpsilon);
}
float ReconstructionFilter::eval_discretized(float x) const {
    const int index = std::min((int) std::fabs(x * m_scale_factor), BF_FILTER_RESOLUTION);
    return m_values[index];
}
bf_rfilter ReconstructionFilter::flatten(uint32_t block_size) const {
    bf_rfilter f;
    std::memset(&f, 0, sizeof(f));
    f.radius = m_radius;
    f.scale = m_scale_factor;
    f.border = m_border_size;
    f.block_size = block_size;
    std::memcpy(f.values, m_values, sizeof(f.values));
    return f;
}

static ref<ReconstructionFilter> find_filter(const Properties &props, const char *def_plugin) {
    for (auto &kv : props.objects(false)) {
        auto *f = dynamic_cast<ReconstructionFilter *>(kv.second.get());
        if (f) {
            props.mark_queried(kv.first);
            return f;
        }
    }
    ref<Object> o = PluginManager::instance()->create_object(Properties(def_plugin), "ReconstructionFilter");
    return dynamic_cast<ReconstructionFilter *>(o.get());
}

Film::Film(const Properties &props) {
    m_full_width = (uint32_t) props.int_("width", 768);         // film.cpp:10-14
    m_full_height = (uint32_t) props.int_("height", 576);
    // crop window, by default the full film (film.cpp:17-27, set_crop_window :54-64: the size does not adjust to the offset)
    const int64_t ox = props.int_("crop_offset_x", 0), oy = props.int_("crop_offset_y", 0);
    const int64_t cw = props.int_("crop_width", m_full_width), ch = props.int_("crop_height", m_full_height);
    if (ox < 0 || oy < 0 || cw <= 0 || ch <= 0 || ox + cw > (int64_t) m_full_width || oy + ch > (int64_t) m_full_height)
        Throw("Invalid crop window specification!\noffset [%lld, %lld] + crop size [%lld, %lld] vs full size [%u, %u]", (long long) ox, (long long) oy,
              (long long) cw, (long long) ch, m_full_width, m_full_height);
    m_width = (uint32_t) cw;
    m_height = (uint32_t) ch;
    m_crop_x = (uint32_t) ox;
    m_crop_y = (uint32_t) oy;
    m_high_quality_edges = props.bool_("high_quality_edges", false);
    m_filter = find_filter(props, "gaussian");
}
void Film::prepare(const std::vector<std::string> &channels) {
    m_channels = channels;
    m_storage.assign((size_t) m_width * m_height * channels.size(), 0.f);
}
void Film::put(const float *data, size_t n) {
    if (n != m_storage.size()) Throw("Film::put(): mismatched block size");
    for (size_t i = 0; i < n; ++i) m_storage[i] += data[i];
}

// MTS_WAVELENGTH_MIN / MAX of the fork at HEAD (spectrum.h:15-30): compile-time constants there
static constexpr float kLambdaMinNm = 7555556.f, kLambdaMaxNm = 9714286.f;

PhasedArray::PhasedArray(const Properties &props) {
    const int n = (int) props.int_("n_elems", 1);
    if (n < 1 || n > 256) Throw("phased array: n_elems must be in [1, 256]");
    Vector3f steer = props.vector3f("steering_vector", Vector3f());
    steer = Vector3f{std::sin(steer.x), std::sin(steer.y), std::sin(steer.z)};          // steer_vec = sin(steer_vec)
    const Transform4f array_to_world = props.transform("array_loc", Transform4f());
    const Vector3f wid = props.vector3f("elem_dims", Vector3f());
    const Vector3f spacing = props.vector3f("elem_spacing", Vector3f());
    const Vector3f axis = props.vector3f("elem_axis", Vector3f());
    elem_dims[0] = wid.x; elem_dims[1] = wid.y; elem_dims[2] = wid.z;
    const Vector3f step{spacing.x * axis.x, spacing.y * axis.y, spacing.z * axis.z};
    std::vector<Vector3f> locs((size_t) n);
    for (int i = 0; i < n; ++i) {
        // array_centre - elem_spacing*elem_axis*(i - (n/2.f) + 0.5)   (even n)   or   (i - (n-1.f)/2.f)   (odd n)
        const float k = (n % 2 == 0) ? (float) ((double) ((float) i - (float) n / 2.f) + 0.5) : (float) i - ((float) n - 1.f) / 2.f;
        locs[(size_t) i] = Vector3f{0.f - step.x * k, 0.f - step.y * k, 0.f - step.z * k};
    }
    const float K = (float) (1.0 / ((double) (kLambdaMaxNm - kLambdaMinNm) * 1e-9 / 2));
    n_velems = (uint32_t) (n * n);
    table.assign((size_t) n_velems * BF_VELEM_FLOATS, 0.f);
    auto mulv = [](const Matrix4f &m, float x, float y, float z) {            // 3x3 part times a vector
        return Vector3f{m.m[0] * x + m.m[1] * y + m.m[2] * z, m.m[4] * x + m.m[5] * y + m.m[6] * z, m.m[8] * x + m.m[9] * y + m.m[10] * z};
    };
    auto normalized = [](Vector3f v) {
        const float inv = 1.f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
        return Vector3f{v.x * inv, v.y * inv, v.z * inv};
    };
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const Vector3f a = locs[(size_t) i], b = locs[(size_t) j];
            const Vector3f r_v{(a.x + b.x) / 2, (a.y + b.y) / 2, (a.z + b.z) / 2};
            const Vector3f r_dash{a.x - b.x, a.y - b.y, a.z - b.z};
            const Transform4f t = array_to_world * Transform4f::translate(r_v) * Transform4f::scale(Vector3f{wid.x / 2, wid.y / 2, wid.z});
            const Vector3f dp_du = mulv(t.matrix, 2.f, 0.f, 0.f), dp_dv = mulv(t.matrix, 0.f, 2.f, 0.f);
            // normals transform with the inverse transpose: column 2 of the inverse's rows
            const Vector3f nrm = normalized(Vector3f{t.inverse.m[8], t.inverse.m[9], t.inverse.m[10]});
            const Vector3f fs = normalized(dp_du), ft = normalized(dp_dv), fn = normalized(nrm);
            float *row = table.data() + ((size_t) i * (size_t) n + (size_t) j) * BF_VELEM_FLOATS;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 4; ++c) row[4 * r + c] = t.inverse.m[4 * r + c];     // m_velem_to_object
            const Vector3f rows[3] = {fs, ft, fn};                                           // Transform::from_frame: rows s, t, n
            for (int r = 0; r < 3; ++r) {
                row[12 + 4 * r + 0] = rows[r].x;
                row[12 + 4 * r + 1] = rows[r].y;
                row[12 + 4 * r + 2] = rows[r].z;
            }
            row[24] = r_dash.x; row[25] = r_dash.y; row[26] = r_dash.z;
            // m_psi_dash = exp(j * K * dot(array_centre - r', steer_vec))
            const float theta = K * ((0.f - r_dash.x) * steer.x + (0.f - r_dash.y) * steer.y + (0.f - r_dash.z) * steer.z);
            row[28] = std::cos(theta);
            row[29] = std::sin(theta);
        }
}
bf_phased_array PhasedArray::flat() const {
    bf_phased_array a{};
    a.velems = table.data();
    a.n_velems = n_velems;
    for (int k = 0; k < 3; ++k) a.elem_dims[k] = elem_dims[k];
    return a;
}

static std::string exr_path(const std::string &dest) {
    if (dest.empty()) Throw("develop(): no destination file set");
    size_t slash = dest.find_last_of('/'), dot = dest.find_last_of('.');
    return (dot == std::string::npos || (slash != std::string::npos && dot < slash)) ? dest + ".exr" : dest;
}
void Film::develop() const {
    if (m_channels.empty()) Throw("develop(): nothing has been rendered yet");
    write_exr(exr_path(m_dest), m_width, m_height, m_channels, m_storage.data());
}
void ADC::develop() const {
    if (m_channels.empty()) Throw("develop(): nothing has been received yet");
    write_exr(exr_path(m_dest), m_window_t, m_window_f, m_channels, m_storage.data());
}

ADC::ADC(const Properties &props) {
    m_t_bins = (uint32_t) props.int_("t_bins", 1024);           // adc.cpp:9-13
    m_f_bins = (uint32_t) props.int_("f_bins", 1024);
    // window, in bins; by default the full ADC (adc.cpp:26-38, set_window :80-91)
    const int64_t wt = props.int_("window_t_bins", m_t_bins), wf = props.int_("window_f_bins", m_f_bins);
    const int64_t ot = props.int_("window_offset_t", 0), of = props.int_("window_offset_f", 0);
    if (ot < 0 || of < 0 || wt <= 0 || wf <= 0 || ot + wt > (int64_t) m_t_bins || of + wf > (int64_t) m_f_bins)
        Throw("Invalid window specification!\noffset [%lld, %lld] + window size [%lld, %lld] vs full size [%u, %u]", (long long) ot, (long long) of,
              (long long) wt, (long long) wf, m_t_bins, m_f_bins);
    m_window_t = (uint32_t) wt;
    m_window_f = (uint32_t) wf;
    m_window_offset_t = (uint32_t) ot;
    m_window_offset_f = (uint32_t) of;
    m_t_bandwidth = props.float_("t_bandwidth", 3.81e-6f);      // adc.cpp:27-29
    m_f_bandwidth = props.float_("f_bandwidth", 250e6f);
    (void) props.bool_("high_quality_edges", false);
    m_filter = find_filter(props, "gaussian");                  // adc.cpp:70-75 default
}
void ADC::prepare(const std::vector<std::string> &channels) {
    m_channels = channels;
    m_storage.assign((size_t) m_window_t * m_window_f * channels.size(), 0.f);      // HDRADC::prepare: SignalBlock(m_window_size) (hdradc.cpp:166)
}
void ADC::put(const float *data, size_t n) {
    if (n != m_storage.size()) Throw("ADC::put(): mismatched block size");
    for (size_t i = 0; i < n; ++i) m_storage[i] += data[i];
}

Endpoint::Endpoint(const Properties &props) { m_to_world = props.transform("to_world", Transform4f()); }

template <typename T> static ref<T> find_child(const Properties &props, const char *what, bool unique = true) {
    ref<T> r;
    for (auto &kv : props.objects(false)) {
        auto *p = dynamic_cast<T *>(kv.second.get());
        if (p) {
            if (r && unique) Throw("Only one %s can be specified per object.", what);
            r = p;
            props.mark_queried(kv.first);
        }
    }
    return r;
}

Sensor::Sensor(const Properties &props) : Endpoint(props) {
    m_shutter_open = props.float_("shutter_open", 0.f);         // sensor.cpp
    m_shutter_open_time = props.float_("shutter_close", 0.f) - m_shutter_open;
    if (m_shutter_open_time < 0) Throw("Shutter opening time must be less than or equal to the shutter closing time!");
    m_film = find_child<Film>(props, "film");
    m_sampler = find_child<Sampler>(props, "sampler");
    auto pm = PluginManager::instance();
    if (!m_film) m_film = dynamic_cast<Film *>(pm->create_object(Properties("hdrfilm"), "Film").get());
    if (!m_sampler) {
        Properties ps("independent");
        ps.set_long("sample_count", 4);
        m_sampler = dynamic_cast<Sampler *>(pm->create_object(ps, "Sampler").get());
    }
}

Receiver::Receiver(const Properties &props) : Endpoint(props) {
    m_adc_sampling_start = props.float_("adc_sampling_start", 0.f);        // receiver.cpp:16-24
    m_adc_sampling_time = props.float_("adc_sampling_end", 0.f) - m_adc_sampling_start;
    m_receive_type = props.string("receive_type", "raw");
    if (m_adc_sampling_time < 0) Throw("ADC sampling time must be less than or equal to the adc sampling end time!");
    m_adc = find_child<ADC>(props, "adc");
    m_sampler = find_child<Sampler>(props, "sampler");
    auto pm = PluginManager::instance();
    if (!m_adc) m_adc = dynamic_cast<ADC *>(pm->create_object(Properties("hdradc"), "ADC").get());
    if (!m_sampler) {
        Properties ps("independent");
        ps.set_long("sample_count", 4);
        m_sampler = dynamic_cast<Sampler *>(pm->create_object(ps, "Sampler").get());
    }
}

Shape::Shape(const Properties &props) {
    m_to_world = props.transform("to_world", Transform4f());
    m_velocity = props.transform("velocity", Transform4f());
    // shape.cpp:38-98
    for (auto &kv : props.objects(false)) {
        Object *o = kv.second.get();
        bool used = true;
        if (auto *e = dynamic_cast<Emitter *>(o)) {
            if (m_emitter) Throw("Only a single Emitter child object can be specified per shape.");
            m_emitter = e;
        } else if (auto *t = dynamic_cast<Transmitter *>(o)) {
            if (m_transmitter) Throw("Only a single Transmitter child object can be specified per shape.");
            m_transmitter = t;
        } else if (auto *s = dynamic_cast<Sensor *>(o)) {
            if (m_sensor) Throw("Only a single Sensor child object can be specified per shape.");
            m_sensor = s;
        } else if (auto *r = dynamic_cast<Receiver *>(o)) {
            if (m_receiver) Throw("Only a single Receiver child object can be specified per shape.");
            m_receiver = r;
        } else if (auto *b = dynamic_cast<BSDF *>(o)) {
            if (m_bsdf) Throw("Only a single BSDF child object can be specified per shape.");
            m_bsdf = b;
        } else {
            used = false;
        }
        if (used) props.mark_queried(kv.first);
    }
    if (!m_bsdf) {
        // shape.cpp:89-98: default diffuse; black if the shape emits / transmits
        Properties pb("diffuse");
        if (m_emitter || m_transmitter) pb.set_float("reflectance", 0.0);
        m_bsdf = dynamic_cast<BSDF *>(PluginManager::instance()->create_object(pb, "BSDF").get());
    }
    if (m_emitter) m_emitter->set_shape(this);
    if (m_transmitter) m_transmitter->set_shape(this);
    if (m_sensor) m_sensor->set_shape(this);
    if (m_receiver) m_receiver->set_shape(this);
}

float Mesh::surface_area() const {
    double a = 0;
    for (size_t f = 0; f < m_faces.size() / 3; ++f) {
        const float *p0 = &m_positions[3 * m_faces[3 * f]], *p1 = &m_positions[3 * m_faces[3 * f + 1]],
                    *p2 = &m_positions[3 * m_faces[3 * f + 2]];
        double e1[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, e2[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
        double c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        a += 0.5 * std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    }
    return (float) a;
}

void Mesh::recompute_vertex_normals() {
    // mesh.cpp:201-249: angle-weighted (Thuermer & Wuethrich), fp32 accumulate
    size_t nv = vertex_count(), nf = primitive_count();
    std::vector<float> nrm(3 * nv, 0.f);
    size_t invalid = 0;
    auto V = [&](uint32_t i, double *o) {
        o[0] = m_positions[3 * i];
        o[1] = m_positions[3 * i + 1];
        o[2] = m_positions[3 * i + 2];
    };
    for (size_t f = 0; f < nf; ++f) {
        uint32_t fi[3] = {m_faces[3 * f], m_faces[3 * f + 1], m_faces[3 * f + 2]};
        double v[3][3];
        for (int k = 0; k < 3; ++k) V(fi[k], v[k]);
        double s0[3], s1[3], n[3];
        for (int k = 0; k < 3; ++k) {
            s0[k] = v[1][k] - v[0][k];
            s1[k] = v[2][k] - v[0][k];
        }
        n[0] = s0[1] * s1[2] - s0[2] * s1[1];
        n[1] = s0[2] * s1[0] - s0[0] * s1[2];
        n[2] = s0[0] * s1[1] - s0[1] * s1[0];
        double l2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
        if (!(l2 > 0)) continue;
        double il = 1.0 / std::sqrt(l2);
        for (int k = 0; k < 3; ++k) n[k] *= il;
        for (int j = 0; j < 3; ++j) {
            double a[3], b[3];
            for (int k = 0; k < 3; ++k) {
                a[k] = v[(j + 1) % 3][k] - v[j][k];
                b[k] = v[(j + 2) % 3][k] - v[j][k];
            }
            double la = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), lb = std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
            double c = (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]) / (la * lb);
            double ang = std::acos(std::max(-1.0, std::min(1.0, c)));
            for (int k = 0; k < 3; ++k) nrm[3 * fi[j] + k] += (float) (n[k] * ang);
        }
    }
    for (size_t i = 0; i < nv; ++i) {
        float *n = &nrm[3 * i];
        float len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (len != 0.f) {
            n[0] /= len; n[1] /= len; n[2] /= len;
        } else {
            n[0] = 1; n[1] = 0; n[2] = 0;
            ++invalid;
        }
    }
    if (invalid) Log(Warn, "computed vertex normals (%zu invalid vertices!)", invalid);
    m_normals.swap(nrm);
}

// ---- scene --------------------------------------------------------------------
struct Scene::Flat {
    const Endpoint *endpoint = nullptr;
    std::vector<bf_shape> shapes;
    std::vector<bf_material> materials;
    std::vector<bf_emitter> emitters;
    bf_scene_desc desc;
    bf_scene *device = nullptr;
    std::vector<bf_scene *> more;      // GPUs 1, 2, ... (device_scenes)
    ~Flat() {
        if (device) bf_scene_destroy(device);
        for (bf_scene *s : more) bf_scene_destroy(s);
    }
};

Scene::Scene(const Properties &props) {
    // scene.cpp:22-71
    for (auto &kv : props.objects()) {
        Object *o = kv.second.get();
        if (auto *s = dynamic_cast<Shape *>(o)) {
            m_shapes.push_back(s);
            if (s->emitter()) m_emitters.push_back(s->emitter());
            if (s->transmitter()) m_transmitters.push_back(s->transmitter());
            if (s->sensor()) m_sensors.push_back(s->sensor());
            if (s->receiver()) m_receivers.push_back(s->receiver());
        } else if (auto *e = dynamic_cast<Emitter *>(o)) {
            m_emitters.push_back(e);
        } else if (auto *t = dynamic_cast<Transmitter *>(o)) {
            m_transmitters.push_back(t);
        } else if (auto *se = dynamic_cast<Sensor *>(o)) {
            m_sensors.push_back(se);
        } else if (auto *r = dynamic_cast<Receiver *>(o)) {
            m_receivers.push_back(r);
        } else if (auto *in = dynamic_cast<Integrator *>(o)) {
            if (m_integrator) Throw("Only one integrator can be specified per scene.");
            m_integrator = in;
        }
    }
    auto pm = PluginManager::instance();
    if (m_sensors.empty() && m_receivers.empty()) {
        // scene.cpp:73-98 synthesises a perspective camera that frames the
        // scene; radar scenes always carry a sensor / receiver
        Throw("Scene: no sensor or receiver specified (the auto-camera of scene.cpp:73-98 is not part of the radar path)");
    }
    if (!m_integrator) {
        Log(Warn, "No integrator found! Instantiating a path tracer..");   // scene.cpp:100-104
        m_integrator = dynamic_cast<Integrator *>(pm->create_object(Properties("path"), "Integrator").get());
    }
}
Scene::~Scene() {}

static void copy16(float *dst, const Matrix4f &m) { std::memcpy(dst, m.m, 16 * sizeof(float)); }

void Scene::flatten(const Endpoint *endpoint) {
    auto fl = std::make_unique<Flat>();
    fl->endpoint = endpoint;
    std::map<const BSDF *, uint32_t> mat_index;
    std::memset(&fl->desc, 0, sizeof(fl->desc));
    bf_sensor &sen = fl->desc.sensor;
    sen.shape = -1;
    bool endpoint_found = false;
    // standalone emitters first keep their scene order; shape emitters follow shape order
    std::vector<std::pair<const Endpoint *, int32_t>> em_slots;   // (endpoint, shape index)
    const bool use_transmitters = dynamic_cast<const Receiver *>(endpoint) != nullptr;
    for (size_t i = 0; i < m_shapes.size(); ++i) {
        Shape *s = m_shapes[i].get();
        bf_shape bs;
        std::memset(&bs, 0, sizeof(bs));
        auto it = mat_index.find(s->bsdf());
        if (it == mat_index.end()) {
            mat_index[s->bsdf()] = (uint32_t) fl->materials.size();
            fl->materials.push_back(s->bsdf()->flatten());
            if (const BSDF *back = s->bsdf()->back()) {
                // twosided with two nested BSDFs: the back side is a table entry of its own (flipped like any twosided entry)
                bf_material b = back->flatten();
                b.twosided = 1;
                b.back_material = 0;
                const size_t front = fl->materials.size() - 1;
                fl->materials.push_back(b);
                fl->materials[front].back_material = (uint32_t) fl->materials.size();      // index of `b`, plus one
            }
            it = mat_index.find(s->bsdf());
        }
        bs.material = it->second;
        bs.emitter = -1;
        copy16(bs.velocity, s->velocity().matrix);
        if (s->is_rectangle()) {
            bs.type = BF_SHAPE_RECTANGLE;
            copy16(bs.to_world, s->to_world().matrix);
            copy16(bs.to_object, s->to_world().inverse);
        } else {
            bs.type = BF_SHAPE_MESH;
            copy16(bs.to_world, Matrix4f::identity());
            copy16(bs.to_object, Matrix4f::identity());
            bs.positions = s->positions()->data();
            bs.normals = s->normals() ? s->normals()->data() : nullptr;
            bs.texcoords = s->texcoords() ? s->texcoords()->data() : nullptr;
            bs.indices = s->faces()->data();
            bs.n_vertices = (uint32_t) (s->positions()->size() / 3);
            bs.n_faces = (uint32_t) (s->faces()->size() / 3);
        }
        if ((const Endpoint *) s->sensor() == endpoint || (const Endpoint *) s->receiver() == endpoint) {
            bs.is_sensor = 1;
            sen.shape = (int32_t) i;
            endpoint_found = true;
        }
        fl->shapes.push_back(bs);
    }
    // emitters in Scene order (scene.cpp:34-61 pushes shape emitters when the shape is visited)
    if (use_transmitters) {
        for (auto &t : m_transmitters) {
            int32_t si = -1;
            for (size_t i = 0; i < m_shapes.size(); ++i)
                if (m_shapes[i]->transmitter() == t.get()) si = (int32_t) i;
            bf_emitter e = t->flatten(si);
            if (si >= 0) fl->shapes[si].emitter = (int32_t) fl->emitters.size();
            fl->emitters.push_back(e);
        }
    } else {
        for (auto &em : m_emitters) {
            int32_t si = -1;
            for (size_t i = 0; i < m_shapes.size(); ++i)
                if (m_shapes[i]->emitter() == em.get()) si = (int32_t) i;
            bf_emitter e = em->flatten(si);
            if (si >= 0) fl->shapes[si].emitter = (int32_t) fl->emitters.size();
            fl->emitters.push_back(e);
        }
    }
    if (auto *se = dynamic_cast<const Sensor *>(endpoint)) {
        se->flatten(sen, sen.shape);
        // render() walks the film in blocks of MTS_BLOCK_SIZE (spiral.h:10; integrator.cpp:101-114 halves it while there are
        // fewer blocks than threads — a choice of the machine, not of the scene: the default is what is flattened)
        uint32_t block = 32;
        if (auto *si = dynamic_cast<const SamplingIntegrator *>(m_integrator.get()))
            if (si->block_size()) block = si->block_size();          // the integrator's "block_size" property
        sen.rfilter = se->film()->reconstruction_filter()->flatten(block);
        sen.crop_offset_x = se->film()->crop_offset_x();
        sen.crop_offset_y = se->film()->crop_offset_y();
        endpoint_found = true;
    } else if (auto *re = dynamic_cast<const Receiver *>(endpoint)) {
        re->flatten(sen, sen.shape);
        sen.rfilter = re->adc()->reconstruction_filter()->flatten(0);      // receive(): one block of the ADC's size
        if (re->adc()->window_t_bins() != re->adc()->t_bins() || re->adc()->window_f_bins() != re->adc()->f_bins()) {
            sen.window_t_bins = re->adc()->window_t_bins();
            sen.window_f_bins = re->adc()->window_f_bins();
            sen.window_offset_t = re->adc()->window_offset_t();
            sen.window_offset_f = re->adc()->window_offset_f();
        }
    }
    if (!endpoint_found) Throw("Scene: the given sensor / receiver does not belong to this scene");
    // physics constants of the fork at HEAD (spectrum.h:15-40, math.h:40-41)
    fl->desc.physics.c = 340.0f;
    fl->desc.physics.lambda_min_nm = kLambdaMinNm;
    fl->desc.physics.lambda_max_nm = kLambdaMaxNm;
    fl->desc.shapes = fl->shapes.data();
    fl->desc.n_shapes = (uint32_t) fl->shapes.size();
    fl->desc.materials = fl->materials.data();
    fl->desc.n_materials = (uint32_t) fl->materials.size();
    fl->desc.emitters = fl->emitters.data();
    fl->desc.n_emitters = (uint32_t) fl->emitters.size();
    m_flat = std::move(fl);      // (the old description's device scenes go with it)
}

const bf_scene_desc *Scene::flat_desc(const Endpoint *endpoint) {
    if (!m_flat || m_flat->endpoint != endpoint) flatten(endpoint);
    return &m_flat->desc;
}

unsigned long long abi_fingerprint_of_host() { return abi_fingerprint_of_this_build(); }

/// libbeifong_hip.so must have been compiled against the header this library was: a core whose bf_stats / bf_launch has
/// another size would write past Integrator::m_stats (the host-side segfault of round 2: a stale libbeifong_host.so
/// against a rebuilt core, DESIGN.md "ABI handshake")
static void check_core_abi() {
    static const bool ok = [] {
        if (bf_version() != BF_ABI_VERSION || bf_abi_fingerprint() != (uint64_t) BF_ABI_FINGERPRINT)
            Throw("libbeifong_hip.so has ABI version %d / layout %016llx, libbeifong_host.so was built for version %d / layout "
                  "%016llx: rebuild both (python -c 'import __graft_entry__ as g; g.build()')",
                  bf_version(), (unsigned long long) bf_abi_fingerprint(), BF_ABI_VERSION, (unsigned long long) BF_ABI_FINGERPRINT);
        return true;
    }();
    (void) ok;
}

bf_scene *Scene::device_scene(const Endpoint *endpoint) {
    check_core_abi();
    flat_desc(endpoint);
    if (!m_flat->device) {
        bf_status st = bf_scene_create(&m_flat->desc, &m_flat->device);
        if (st != BF_OK) Throw("bf_scene_create failed (status %d): %s", st, bf_last_error());
    }
    return m_flat->device;
}

std::vector<bf_scene *> Scene::device_scenes(const Endpoint *endpoint, int n) {
    std::vector<bf_scene *> r = {device_scene(endpoint)};
    if (n > bf_device_count()) Throw("%d GPUs requested, %d visible", n, bf_device_count());
    bf_scene_info info;
    if (bf_scene_get_info(r[0], &info) != BF_OK) Throw("bf_scene_get_info: %s", bf_last_error());
    int home = info.device;
    for (int g = 1; g < n; ++g) {
        if ((int) m_flat->more.size() < g) {
            // GPU 0 is wherever the first scene lives; the others are the remaining devices in order
            int dev = g <= home ? g - 1 : g;
            if (bf_set_device(dev) != BF_OK) Throw("bf_set_device(%d): %s", dev, bf_last_error());
            bf_scene *s = nullptr;
            bf_status st = bf_scene_create(&m_flat->desc, &s);
            (void) bf_set_device(home);
            if (st != BF_OK) Throw("bf_scene_create on GPU %d failed (status %d): %s", dev, st, bf_last_error());
            m_flat->more.push_back(s);
        }
        r.push_back(m_flat->more[g - 1]);
    }
    return r;
}

/// one render on gpu_count() GPUs: the plain entry for one, sample shards + RCCL all-reduce for more (bf_render_sharded)
static bf_status render_on_gpus(Scene *scene, const Endpoint *endpoint, const bf_launch &lp, float *hist, bf_stats *stats) {
    const int n = gpu_count();
    if (n <= 1) return bf_render(scene->device_scene(endpoint), &lp, hist, nullptr, stats);
    std::vector<bf_scene *> scenes = scene->device_scenes(endpoint, n);
    return bf_render_sharded(scenes.data(), (uint32_t) scenes.size(), &lp, hist, stats);
}

// ---- integrator ------------------------------------------------------------------
SamplingIntegrator::SamplingIntegrator(const Properties &props) : Integrator(props) {
    // SamplingIntegrator — integrator.cpp:26-43
    const int64_t bs = props.int_("block_size", 0);
    if (bs < 0) Throw("\"block_size\" must not be negative");
    m_block_size = (uint32_t) bs;
    uint32_t pow2 = 1;
    while (pow2 < m_block_size) pow2 <<= 1;
    if (m_block_size > 0 && pow2 != m_block_size) {
        Log(Warn, "Setting block size from %u to next higher power of two: %u", m_block_size, pow2);
        m_block_size = pow2;
    }
    const int64_t spp_pass = props.int_("samples_per_pass", -1);
    m_samples_per_pass = spp_pass < 0 ? (size_t) -1 : (size_t) spp_pass;
    (void) props.float_("timeout", -1.f);            // a render is milliseconds of GPU work: nothing to time out
    (void) props.bool_("hide_emitters", false);      // read by `direct` / `volpath*` only, never by the radar integrators
    // MonteCarloIntegrator — integrator.cpp:1713-1728
    m_rr_depth = (int) props.int_("rr_depth", 5);
    if (m_rr_depth <= 0) Throw("\"rr_depth\" must be set to a value greater than zero!");
    m_max_depth = (int) props.int_("max_depth", -1);
    if (m_max_depth < 0 && m_max_depth != -1) Throw("\"max_depth\" must be set to -1 (infinite) or a value >= 0");
    // not a reference property: switches on Shape::doppler at the three call sites the reference carries commented
    // out (pathtimefrequency.cpp:124-126,141-144,180-183); false = the reference's HEAD
    m_doppler = props.bool_("doppler", false);
}

static uint32_t color_mode_of_variant() {
    // scalar_rgb converts through srgb_to_xyz (integrator.cpp:292-294); mono and
    // spectral-with-uniform-spectra replicate the lane (CIE tables are out of scope)
    return variant() == "scalar_rgb" ? BF_COLOR_RGB : BF_COLOR_MONO;
}

bool SamplingIntegrator::render(Scene *scene, Sensor *sensor) {
    // integrator.cpp:58-204: channels X,Y,Z,A,W + aov_names() per pixel
    Film *film = sensor->film();
    std::vector<std::string> channels = {"X", "Y", "Z", "A", "W"};
    for (auto &n : aov_names()) channels.push_back(n);
    film->prepare(channels);
    bf_launch lp;
    std::memset(&lp, 0, sizeof(lp));
    lp.color_mode = color_mode_of_variant();
    lp.n_paths = sensor->sampler()->sample_count();
    {
        // integrator.cpp:66-75: passes of samples_per_pass samples; all of them are one launch here
        const size_t total = sensor->sampler()->sample_count(), per_pass = m_samples_per_pass == (size_t) -1 ? total : std::min(m_samples_per_pass, total);
        if (per_pass == 0 || total % per_pass != 0)
            Throw("sample_count (%zu) must be a multiple of samples_per_pass (%zu).", total, per_pass);
    }
    if (film->width() != 1 || film->height() != 1) {
        // sample_count samples per pixel, pixels row-major (bf_launch.spp)
        if (lp.n_paths > 0xffffffffull) Throw("sample_count %llu is too large", (unsigned long long) lp.n_paths);
        lp.film_width = film->width();
        lp.film_height = film->height();
        lp.spp = (uint32_t) lp.n_paths;
        lp.n_paths *= (uint64_t) film->width() * film->height();
    }
    lp.seed = sensor->sampler()->base_seed();
    lp.max_depth = max_depth();
    lp.rr_depth = rr_depth();
    lp.time_c = 3.0e8f;
    configure(lp);
    if (lp.mode == BF_MODE_RECEIVE_RAW) Throw("this integrator only supports receive(), not render()");
    uint32_t n = bf_launch_channels(&lp);
    if (n != channels.size() * film->width() * film->height())
        Throw("internal error: channel count mismatch (%u vs %zu)", n, channels.size() * film->width() * film->height());
    std::vector<float> hist(n);
    auto t0 = std::chrono::steady_clock::now();
    bf_status st = render_on_gpus(scene, sensor, lp, hist.data(), &m_stats.stats);
    if (st != BF_OK) Throw("bf_render failed (status %d): %s", st, bf_last_error());
    m_stats.wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    film->put(hist.data(), hist.size());
    return !m_stop;
}

void SamplingIntegrator::receive_launch(const Receiver *receiver, bf_launch &lp) const {
    const std::string &rt = receiver->receive_type();
    if (rt != "raw" && rt != "raw_resample" && rt != "mix_resample")
        Throw("receive_type \"%s\" is not supported (\"raw\", \"raw_resample\" and \"mix_resample\" are)", rt.c_str());
    std::memset(&lp, 0, sizeof(lp));
    if (rt == "mix_resample") lp.flags |= BF_FLAG_MIX_RESAMPLE;
    if (doppler()) lp.flags |= BF_FLAG_DOPPLER;
    lp.color_mode = BF_COLOR_MONO;
    lp.n_paths = receiver->sampler()->sample_count();
    lp.seed = receiver->sampler()->base_seed();
    lp.max_depth = max_depth();
    lp.rr_depth = rr_depth();
    lp.time_c = 3.0e8f;
    configure(lp);
    if (lp.mode != BF_MODE_RECEIVE_RAW) Throw("this integrator does not implement receive()");
    lp.bins = receiver->adc()->window_t_bins();
    lp.bins_y = receiver->adc()->window_f_bins();
}

bool SamplingIntegrator::receive(Scene *scene, Receiver *receiver) {
    // integrator.cpp:315-768 (live branch :484-666): channels Y,A,W + aov_names()
    ADC *adc = receiver->adc();
    std::vector<std::string> channels = {"Y", "A", "W"};
    for (auto &n : aov_names()) channels.push_back(n);
    adc->prepare(channels);
    // "raw" and "raw_resample" take the same branches everywhere at HEAD (integrator.cpp:1603-1623,
    // wignerreceiver.cpp:64-71,174-178).  "mix_resample" bins the beat frequency |f_after - f_rx| (:1588-1603), which is
    // exactly 0 while the Doppler update is commented out (pathtimefrequency.cpp:440-445) and so lands outside the ADC
    // (SignalBlock::put: ceil(0 - 1) = -1): BF_FLAG_MIX_RESAMPLE reproduces that.  "mixer" is an empty branch that would
    // bin an uninitialised coordinate (:1624-1634): rejected.
    bf_launch lp;
    receive_launch(receiver, lp);
    uint32_t n = bf_launch_channels(&lp);
    std::vector<float> hist(n);
    auto t0 = std::chrono::steady_clock::now();
    bf_status st = render_on_gpus(scene, receiver, lp, hist.data(), &m_stats.stats);
    if (st != BF_OK) Throw("bf_render failed (status %d): %s", st, bf_last_error());
    m_stats.wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    adc->put(hist.data(), hist.size());
    return !m_stop;
}

}  // namespace bfh
