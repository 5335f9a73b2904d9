// src/films/hdrfilm.cpp — HDRFilm (storage + raw bitmap only; EXR output is out of scope)
#include "../render.h"
using namespace bfh;
class HDRFilm final : public Film {
public:
    explicit HDRFilm(const Properties &props) : Film(props) {
        (void) props.string("file_format", "openexr");
        (void) props.string("pixel_format", "rgba");
        (void) props.string("component_format", "float16");
        (void) props.bool_("high_quality_edges", false);
    }
};
BF_EXPORT_PLUGIN(HDRFilm, "Film", "hdrfilm", "HDR Film")
