// TEST-ONLY stand-in, see Foundation/Foundation.h in this directory.
#pragma once
#import <Metal/Metal.h>
