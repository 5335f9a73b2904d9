// src/adcs/hdradc.cpp:101-150 — HDRADC (fork): raw storage + bitmap, multi-channel EXR output; the property checks of the constructor
#include <algorithm>
#include "../render.h"
using namespace bfh;
static std::string lower(std::string s) {
    std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char) std::tolower(c); });
    return s;
}
class HDRADC final : public ADC {
public:
    explicit HDRADC(const Properties &props) : ADC(props) {
        const std::string file_format = lower(props.string("file_format", "openexr"));
        const std::string pixel_format = lower(props.string("pixel_format", "luminance"));
        const std::string component_format = lower(props.string("component_format", "float16"));
        const std::string filename = props.string("filename", "");
        if (!filename.empty()) set_destination_file(filename);
        if (file_format != "openexr" && file_format != "exr")
            Throw("The \"file_format\" parameter must be equal to \"openexr\", found %s instead.", file_format.c_str());
        // (the monochromatic variants accept any value with a warning, hdradc.cpp:125-130; the radar variants are not monochromatic)
        if (pixel_format != "luminance") Throw("The \"pixel_format\" parameter must be equal to \"luminance\". Found %s.", pixel_format.c_str());
        if (component_format != "float16" && component_format != "float32" && component_format != "uint32")
            Throw("The \"component_format\" parameter must either be equal to \"float16\", \"float32\", or \"uint32\". Found %s instead.",
                  component_format.c_str());
        props.mark_queried("banner");      // no banner in Mitsuba 2
    }
};
BF_EXPORT_PLUGIN(HDRADC, "ADC", "hdradc", "HDR ADC")
