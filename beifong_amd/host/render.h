// beifong_amd host layer — render classes.
//
// Source-level mirror of the librender surface the radar hot path touches
// (include/mitsuba/render/*.h): Scene, Shape/Mesh, BSDF, Endpoint -> {Emitter,
// Sensor, Transmitter, Receiver}, Film/ADC, Sampler, ReconstructionFilter,
// Texture (constant spectra only) and Integrator/SamplingIntegrator with both
// entry points render(Scene*, Sensor*) and receive(Scene*, Receiver*)
// (integrator.h:43-53).  The objects only HOLD the scene description; the
// integrator flattens it into the C ABI's bf_scene_desc and the HIP library
// does the work.
#pragma once
#include "../../include/beifong_hip.h"
#include "core.h"

namespace bfh {

class Scene;

/// constant spectrum (the only textures radar scenes use: uniform / 2-point regular)
class Texture : public Object {
public:
    explicit Texture(float v) : m_value(v) {}
    float value() const { return m_value; }
    const Class *class_() const override;

private:
    float m_value;
};

/// include/mitsuba/core/rfilter.h:48-82, src/libcore/rfilter.cpp:9-21
class ReconstructionFilter : public Object {
public:
    float radius() const { return m_radius; }
    uint32_t border_size() const { return m_border_size; }
    virtual float eval(float x) const = 0;
    /// rfilter.h:62-65
    float eval_discretized(float x) const;
    /// the discretised filter as the C ABI carries it (bf_rfilter); block_size: edge of the blocks the film is rendered in
    bf_rfilter flatten(uint32_t block_size) const;
    const Class *class_() const override;

protected:
    void init_discretization();
    float m_radius = 0.f, m_scale_factor = 0.f;
    float m_values[BF_FILTER_RESOLUTION + 1] = {};
    uint32_t m_border_size = 0;
};

class Sampler : public Object {
public:
    explicit Sampler(const Properties &props);
    size_t sample_count() const { return m_sample_count; }
    uint64_t base_seed() const { return m_base_seed; }
    const Class *class_() const override;

protected:
    size_t m_sample_count;
    uint64_t m_base_seed;
};

class BSDF : public Object {
public:
    /// flatten into the material table entry (diffuse.cpp, roughconductor.cpp, twosided.cpp)
    virtual bf_material flatten() const = 0;
    /// TwoSidedBRDF with two nested BSDFs: the one that shades the back side (else null)
    virtual const BSDF *back() const { return nullptr; }
    const Class *class_() const override;
};

class Film : public Object {
public:
    explicit Film(const Properties &props);
    /// crop_size(): what is rendered and stored (= size() without a crop window)
    uint32_t width() const { return m_width; }
    uint32_t height() const { return m_height; }
    /// Film::size() / crop_offset() / has_high_quality_edges() — film.cpp:10-27
    uint32_t full_width() const { return m_full_width; }
    uint32_t full_height() const { return m_full_height; }
    uint32_t crop_offset_x() const { return m_crop_x; }
    uint32_t crop_offset_y() const { return m_crop_y; }
    bool has_high_quality_edges() const { return m_high_quality_edges; }
    const ReconstructionFilter *reconstruction_filter() const { return m_filter.get(); }
    /// HDRFilm::prepare / put / bitmap(raw) — hdrfilm.cpp:190-211,251-275
    void prepare(const std::vector<std::string> &channels);
    void put(const float *data, size_t n);
    const std::vector<float> &bitmap() const { return m_storage; }
    const std::vector<std::string> &channels() const { return m_channels; }
    /// HDRFilm::set_destination_file / develop — hdrfilm.cpp:213-249 (".exr" is appended if there is no extension)
    void set_destination_file(const std::string &path) { m_dest = path; }
    void develop() const;
    const Class *class_() const override;

protected:
    std::string m_dest;
    uint32_t m_width, m_height, m_full_width, m_full_height, m_crop_x, m_crop_y;
    bool m_high_quality_edges = false;
    ref<ReconstructionFilter> m_filter;
    std::vector<std::string> m_channels;
    std::vector<float> m_storage;    // [H][W][C]
};

/// src/librender/adc.cpp:7-91, src/adcs/hdradc.cpp
class ADC : public Object {
public:
    explicit ADC(const Properties &props);
    uint32_t t_bins() const { return m_t_bins; }
    uint32_t f_bins() const { return m_f_bins; }
    /// ADC::window_size / window_offset (adc.h): what receive() bins into and the storage holds
    uint32_t window_t_bins() const { return m_window_t; }
    uint32_t window_f_bins() const { return m_window_f; }
    uint32_t window_offset_t() const { return m_window_offset_t; }
    uint32_t window_offset_f() const { return m_window_offset_f; }
    float t_bandwidth() const { return m_t_bandwidth; }
    float f_bandwidth() const { return m_f_bandwidth; }
    const ReconstructionFilter *reconstruction_filter() const { return m_filter.get(); }
    void prepare(const std::vector<std::string> &channels);
    void put(const float *data, size_t n);
    const std::vector<float> &bitmap() const { return m_storage; }   // [f][t][C]
    const std::vector<std::string> &channels() const { return m_channels; }
    /// HDRADC::set_destination_file / develop — hdradc.cpp:259-295
    void set_destination_file(const std::string &path) { m_dest = path; }
    void develop() const;
    const Class *class_() const override;

protected:
    uint32_t m_t_bins, m_f_bins;
    uint32_t m_window_t, m_window_f, m_window_offset_t, m_window_offset_f;
    float m_t_bandwidth, m_f_bandwidth;
    ref<ReconstructionFilter> m_filter;
    std::vector<std::string> m_channels;
    std::vector<float> m_storage;
    std::string m_dest;
};

class Shape;
class Endpoint : public Object {
public:
    explicit Endpoint(const Properties &props);
    void set_shape(Shape *s) { m_shape = s; }
    Shape *shape() const { return m_shape; }
    const Transform4f &world_transform() const { return m_to_world; }
    const Class *class_() const override;

protected:
    Transform4f m_to_world;
    Shape *m_shape = nullptr;
};

class Emitter : public Endpoint {
public:
    using Endpoint::Endpoint;
    virtual bf_emitter flatten(int32_t shape_index) const = 0;
    const Class *class_() const override;
};
/// fork: include/mitsuba/render/transmitter.h
/// The n_elems^2 virtual elements the phased-array constructors precompute (phasedtransmitter.cpp:108-165 ==
/// phasedreceiver.cpp:115-172), in the layout of bf_phased_array (BF_VELEM_FLOATS floats per element).  Reads the
/// properties n_elems, steering_vector, array_loc, elem_dims, elem_spacing, elem_axis.
struct PhasedArray {
    std::vector<float> table;
    float elem_dims[3] = {0, 0, 0};
    uint32_t n_velems = 0;
    explicit PhasedArray(const Properties &props);
    bf_phased_array flat() const;
};

class Transmitter : public Endpoint {
public:
    using Endpoint::Endpoint;
    virtual bf_emitter flatten(int32_t shape_index) const = 0;
    const Class *class_() const override;
};

class Sensor : public Endpoint {
public:
    explicit Sensor(const Properties &props);
    Film *film() const { return m_film.get(); }
    Sampler *sampler() const { return m_sampler.get(); }
    virtual void flatten(bf_sensor &out, int32_t shape_index) const = 0;
    const Class *class_() const override;

protected:
    ref<Film> m_film;
    ref<Sampler> m_sampler;
    float m_shutter_open, m_shutter_open_time;
};
/// fork: src/librender/receiver.cpp:16-62
class Receiver : public Endpoint {
public:
    explicit Receiver(const Properties &props);
    ADC *adc() const { return m_adc.get(); }
    Sampler *sampler() const { return m_sampler.get(); }
    const std::string &receive_type() const { return m_receive_type; }
    virtual void flatten(bf_sensor &out, int32_t shape_index) const = 0;
    const Class *class_() const override;

protected:
    ref<ADC> m_adc;
    ref<Sampler> m_sampler;
    float m_adc_sampling_start, m_adc_sampling_time;
    std::string m_receive_type;
};

class Shape : public Object {
public:
    explicit Shape(const Properties &props);      // shape.cpp:38-98: sorts children into slots
    BSDF *bsdf() const { return m_bsdf.get(); }
    Emitter *emitter() const { return m_emitter.get(); }
    Transmitter *transmitter() const { return m_transmitter.get(); }
    Sensor *sensor() const { return m_sensor.get(); }
    Receiver *receiver() const { return m_receiver.get(); }
    const Transform4f &to_world() const { return m_to_world; }
    /// "velocity": the transform Shape::doppler applies (src/librender/shape.cpp:42,375-389), default identity
    const Transform4f &velocity() const { return m_velocity; }
    virtual uint32_t primitive_count() const = 0;
    virtual float surface_area() const = 0;
    virtual bool is_rectangle() const { return false; }
    /// mesh arrays (world space) or nullptr for analytic shapes
    virtual const std::vector<float> *positions() const { return nullptr; }
    virtual const std::vector<float> *normals() const { return nullptr; }
    virtual const std::vector<float> *texcoords() const { return nullptr; }
    virtual const std::vector<uint32_t> *faces() const { return nullptr; }
    const Class *class_() const override;

protected:
    Transform4f m_to_world;
    Transform4f m_velocity;
    ref<BSDF> m_bsdf;
    ref<Emitter> m_emitter;
    ref<Transmitter> m_transmitter;
    ref<Sensor> m_sensor;
    ref<Receiver> m_receiver;
};

/// include/mitsuba/render/mesh.h — vertex / face buffers, world space
class Mesh : public Shape {
public:
    using Shape::Shape;
    uint32_t primitive_count() const override { return (uint32_t) (m_faces.size() / 3); }
    uint32_t vertex_count() const { return (uint32_t) (m_positions.size() / 3); }
    float surface_area() const override;
    const std::vector<float> *positions() const override { return &m_positions; }
    const std::vector<float> *normals() const override { return m_normals.empty() ? nullptr : &m_normals; }
    const std::vector<float> *texcoords() const override { return m_texcoords.empty() ? nullptr : &m_texcoords; }
    const std::vector<uint32_t> *faces() const override { return &m_faces; }
    bool has_vertex_normals() const { return !m_normals.empty(); }
    bool has_vertex_texcoords() const { return !m_texcoords.empty(); }
    const Class *class_() const override;

protected:
    /// Mesh::recompute_vertex_normals — mesh.cpp:201-278 (angle-weighted)
    void recompute_vertex_normals();
    std::vector<float> m_positions, m_normals, m_texcoords;
    std::vector<uint32_t> m_faces;
};

struct RenderStats {
    bf_stats stats;
    double wall_ms;
};

class Integrator : public Object {
public:
    explicit Integrator(const Properties &props) { (void) props; }
    virtual bool render(Scene *scene, Sensor *sensor) = 0;        // integrator.h:43
    virtual bool receive(Scene *scene, Receiver *receiver) = 0;   // integrator.h:44 (fork)
    virtual void cancel() { m_stop = true; }
    const RenderStats &last_stats() const { return m_stats; }
    const Class *class_() const override;

protected:
    bool m_stop = false;
    RenderStats m_stats{};
};

/// integrator.h:116-158.  `sample()` lives in the HIP kernels; subclasses only
/// say which estimator they are and what AOVs they add.
class SamplingIntegrator : public Integrator {
public:
    explicit SamplingIntegrator(const Properties &props);
    bool render(Scene *scene, Sensor *sensor) override;
    bool receive(Scene *scene, Receiver *receiver) override;
    virtual std::vector<std::string> aov_names() const { return {}; }
    /// BF_MODE_* + binning parameters for bf_launch
    virtual void configure(bf_launch &launch) const = 0;
    /// MonteCarloIntegrator parameters; AOV wrappers (range, time, phase) forward to their sub-integrator
    virtual int max_depth() const { return m_max_depth; }
    virtual int rr_depth() const { return m_rr_depth; }
    /// "doppler" (not a reference property, default false = the reference's HEAD): BF_FLAG_DOPPLER for receive()
    virtual bool doppler() const { return m_doppler; }
    /// SamplingIntegrator properties (integrator.cpp:27-43): edge of the image blocks (0: MTS_BLOCK_SIZE) and samples per pass
    uint32_t block_size() const { return m_block_size; }
    size_t samples_per_pass() const { return m_samples_per_pass; }
    /// the bf_launch receive() hands to bf_render for this receiver (receive_type -> flags, ADC -> bins)
    void receive_launch(const Receiver *receiver, bf_launch &launch) const;
    const Class *class_() const override;

protected:
    int m_max_depth = -1, m_rr_depth = 5;
    bool m_doppler = false;
    uint32_t m_block_size = 0;
    size_t m_samples_per_pass = (size_t) -1;
};

/// src/librender/scene.cpp:22-120
class Scene : public Object {
public:
    explicit Scene(const Properties &props);
    ~Scene() override;
    const std::vector<ref<Shape>> &shapes() const { return m_shapes; }
    const std::vector<ref<Sensor>> &sensors() const { return m_sensors; }
    const std::vector<ref<Receiver>> &receivers() const { return m_receivers; }
    const std::vector<ref<Emitter>> &emitters() const { return m_emitters; }
    const std::vector<ref<Transmitter>> &transmitters() const { return m_transmitters; }
    Integrator *integrator() const { return m_integrator.get(); }
    /// flattened description + device scene for the given endpoint (cached)
    bf_scene *device_scene(const Endpoint *endpoint);
    /// one device scene per GPU 0 .. n - 1 (cached; [0] is device_scene()): the handles bf_render_sharded takes
    std::vector<bf_scene *> device_scenes(const Endpoint *endpoint, int n);
    const bf_scene_desc *flat_desc(const Endpoint *endpoint);
    const Class *class_() const override;

private:
    struct Flat;
    void flatten(const Endpoint *endpoint);
    std::vector<ref<Shape>> m_shapes;
    std::vector<ref<Sensor>> m_sensors;
    std::vector<ref<Receiver>> m_receivers;
    std::vector<ref<Emitter>> m_emitters;
    std::vector<ref<Transmitter>> m_transmitters;
    ref<Integrator> m_integrator;
    std::unique_ptr<Flat> m_flat;
};

// xml.cpp
namespace xml {
using ParameterList = std::vector<std::pair<std::string, std::string>>;
ref<Object> load_file(const std::string &filename, const ParameterList &params = {});
ref<Object> load_string(const std::string &xml, const ParameterList &params = {}, const std::string &base_dir = ".");
}  // namespace xml

inline unsigned long long host_class_layout_inline() {
    return (unsigned long long) sizeof(Integrator) << 40 ^ (unsigned long long) sizeof(Scene) << 20 ^ (unsigned long long) sizeof(RenderStats);
}
inline unsigned long long abi_fingerprint_of_this_build() { return (unsigned long long) BF_ABI_FINGERPRINT ^ (host_class_layout_inline() * 0x9e3779b97f4a7c15ull); }

/// file lookup relative to the scene file (FileResolver)
std::string resolve_path(const std::string &path);
void push_search_path(const std::string &dir);

}  // namespace bfh
