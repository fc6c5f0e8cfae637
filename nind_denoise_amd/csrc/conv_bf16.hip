// bf16 instantiations of the convolution kernel (see conv_qp.inc)
#include "conv_qp.inc"
extern const Variant g_variants_bf16[kGroup] = {ND_VARIANT_GROUP(ND_BF16, "bf16")};
