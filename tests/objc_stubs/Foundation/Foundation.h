// TEST-ONLY stand-in (tests/test_host_cpu.py::test_objc_binding_parses): the handful of Foundation declarations
// objc/MetalBT709Decoder+HIP.m and the reference's Renderer/MetalBT709Decoder.h / MetalRenderContext.h use, so that ROCm
// clang can run -fsyntax-only -fobjc-arc over the binding on a machine with no Apple SDK.  Nothing here is built,
// linked or shipped; it makes no reference build.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
typedef signed char BOOL;
#define YES ((BOOL)1)
#define NO ((BOOL)0)
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif
#ifndef nil
#define nil ((void *)0)
#endif
typedef unsigned long NSUInteger;
typedef long NSInteger;
typedef double CGFloat;
typedef struct { CGFloat width, height; } CGSize;
__attribute__((objc_root_class))
@interface NSObject
+ (instancetype)alloc;
- (instancetype)init;
@end
@interface NSString : NSObject
@end
@interface NSArray<ObjectType> : NSObject
@end
@interface NSData : NSObject
@end
extern void NSLog(NSString *format, ...);
