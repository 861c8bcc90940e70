// TEST-ONLY stand-in, see Foundation/Foundation.h in this directory.
#pragma once
#import <Foundation/Foundation.h>
typedef const void *CFTypeRef;
typedef const struct __CFString *CFStringRef;
typedef unsigned char Boolean;
extern Boolean CFEqual(CFTypeRef a, CFTypeRef b);
typedef struct __CVBuffer *CVBufferRef;
typedef CVBufferRef CVImageBufferRef;
typedef CVImageBufferRef CVPixelBufferRef;
typedef uint32_t CVAttachmentMode;
typedef int32_t CVReturn;
typedef uint64_t CVPixelBufferLockFlags;
enum { kCVPixelBufferLock_ReadOnly = 1 };
extern const CFStringRef kCVImageBufferYCbCrMatrixKey, kCVImageBufferYCbCrMatrix_ITU_R_709_2, kCVImageBufferYCbCrMatrix_ITU_R_601_4;
extern const CFStringRef kCVImageBufferTransferFunctionKey, kCVImageBufferTransferFunction_ITU_R_709_2,
    kCVImageBufferTransferFunction_sRGB, kCVImageBufferTransferFunction_Linear;
extern CFTypeRef CVBufferGetAttachment(CVBufferRef buffer, CFStringRef key, CVAttachmentMode *mode);
extern size_t CVPixelBufferGetWidth(CVPixelBufferRef pb);
extern size_t CVPixelBufferGetHeight(CVPixelBufferRef pb);
extern void *CVPixelBufferGetBaseAddressOfPlane(CVPixelBufferRef pb, size_t plane);
extern size_t CVPixelBufferGetBytesPerRowOfPlane(CVPixelBufferRef pb, size_t plane);
extern size_t CVPixelBufferGetHeightOfPlane(CVPixelBufferRef pb, size_t plane);
extern CVReturn CVPixelBufferLockBaseAddress(CVPixelBufferRef pb, CVPixelBufferLockFlags flags);
extern CVReturn CVPixelBufferUnlockBaseAddress(CVPixelBufferRef pb, CVPixelBufferLockFlags flags);
