// f16 instantiations of the convolution kernel (see conv_qp.inc)
#include "conv_qp.inc"
extern const Variant g_variants_f16[kGroup] = {ND_VARIANT_GROUP(ND_F16, "f16")};
