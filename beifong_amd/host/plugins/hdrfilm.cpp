// src/films/hdrfilm.cpp:100-172 — HDRFilm (raw storage + bitmap, multi-channel EXR output; the property checks of the constructor)
#include <algorithm>
#include "../render.h"
using namespace bfh;
static std::string lower(std::string s) {
    std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char) std::tolower(c); });
    return s;
}
class HDRFilm final : public Film {
public:
    explicit HDRFilm(const Properties &props) : Film(props) {
        const std::string file_format = lower(props.string("file_format", "openexr"));
        const std::string pixel_format = lower(props.string("pixel_format", "rgba"));
        const std::string component_format = lower(props.string("component_format", "float16"));
        const std::string filename = props.string("filename", "");
        if (!filename.empty()) set_destination_file(filename);
        if (file_format != "openexr" && file_format != "exr" && file_format != "rgbe" && file_format != "pfm")
            Throw("The \"file_format\" parameter must either be equal to \"openexr\", \"pfm\", or \"rgbe\", found %s instead.", file_format.c_str());
        if (file_format == "rgbe" || file_format == "pfm")
            Log(Warn, "hdrfilm: file_format \"%s\": develop() writes multi-channel OpenEXR (the radar path's only output format)", file_format.c_str());
        static const char *pf[] = {"luminance", "luminance_alpha", "rgb", "rgba", "xyz", "xyza"};
        if (std::find_if(std::begin(pf), std::end(pf), [&](const char *p) { return pixel_format == p; }) == std::end(pf))
            Throw("The \"pixel_format\" parameter must either be equal to \"luminance\", \"luminance_alpha\", \"rgb\", \"rgba\",  \"xyz\", \"xyza\". "
                  "Found %s.", pixel_format.c_str());
        if (component_format != "float16" && component_format != "float32" && component_format != "uint32")
            Throw("The \"component_format\" parameter must either be equal to \"float16\", \"float32\", or \"uint32\". Found %s instead.",
                  component_format.c_str());
        props.mark_queried("banner");      // no banner in Mitsuba 2
    }
};
BF_EXPORT_PLUGIN(HDRFilm, "Film", "hdrfilm", "HDR Film")
