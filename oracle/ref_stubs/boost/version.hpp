#pragma once
#define BOOST_VERSION 107400
