// image_codec.h -- from_ros_to_cv_image (uvo_libraries/src/math_utility.cpp:154-173) without cv_bridge / OpenCV: the payload
// of a sensor_msgs/CompressedImage decoded on the MI355X (uvo_decode_image: host entropy decoding, device IDCT / upsampling /
// colour conversion / demosaicing, byte-identical to libjpeg's defaults), returned as the BGR cv::Mat cv_bridge would produce.
#pragma once
#include <string>
#include "uvo_libraries_hip/VO_utility_hip.h"

namespace uvo_hip {

// data / size: CompressedImage.data; format: CompressedImage.format ("bgr8; jpeg compressed bgr8", "bayer_bggr8; jpeg compressed ...").
// Returns CV_8UC3 (BGR) for colour and bayer messages, CV_8UC1 for mono ones, CV_8UC4 (BGRA) for RGBA PNGs.  Throws
// uvo_hip::Error on payloads the decoder refuses (progressive / arithmetic JPEG, interlaced or 16-bit PNG).  The size query
// parses the headers only.
inline uvocv::Mat decode_compressed_image(const unsigned char* data, size_t size, const std::string& format)
{
    uvo_ctx* c = context();
    int w = 0, h = 0, ch = 0;
    uvo_status st = uvo_decode_image(c, data, size, format.c_str(), nullptr, 0, UVO_MEM_HOST, &w, &h, &ch);
    if (st != UVO_OK) throw Error(st, std::string("uvo_decode_image: ") + uvo_last_error(c));
    uvocv::Mat img(h, w, ch == 4 ? uvocv::CV_8UC4 : (ch == 3 ? uvocv::CV_8UC3 : uvocv::CV_8UC1));
    st = uvo_decode_image(c, data, size, format.c_str(), img.ptr<unsigned char>(0), (size_t)w * h * ch, UVO_MEM_HOST, &w, &h, &ch);
    if (st != UVO_OK) throw Error(st, std::string("uvo_decode_image: ") + uvo_last_error(c));
    return img;
}

}  // namespace uvo_hip
