// src/adcs/hdradc.cpp:101-176 — HDRADC (fork)
#include "../render.h"
using namespace bfh;
class HDRADC final : public ADC {
public:
    explicit HDRADC(const Properties &props) : ADC(props) {
        (void) props.string("file_format", "openexr");
        (void) props.string("pixel_format", "luminance");
        (void) props.string("component_format", "float32");
    }
};
BF_EXPORT_PLUGIN(HDRADC, "ADC", "hdradc", "HDR ADC")
