-- An equal-arc-length lens: the angle is the integral of a speed function, integrated numerically per pixel by a helper that takes
-- the function to integrate as an argument.
local function midpoint(f, a, b, n)
   local h = (b - a) / n
   local acc = 0
   for i = 1, n do acc = acc + f(a + (i - 0.5) * h) * h end
   return acc
end
local function speed(t) return 1 + 0.35 * t * t end
local function largest(first, ...)
   local m = first
   for i = 1, select("#", ...) do
      local v = select(i, ...)
      if v > m then m = v end
   end
   return m
end

max_fov = 240
max_vfov = 240
lens_width = 3
lens_height = 3
onload = "f_contain"

function lens_inverse(x, y)
   local r = sqrt(x * x + y * y)
   local function inside() return largest(abs(x), abs(y)) <= 1.5 end        -- a function defined here, a vararg helper
   if not inside() then return nil end
   if r == 0 then return 0, 0, 1 end
   local theta = midpoint(speed, 0, r, 8) * 0.9 + midpoint(cos, 0, r, 4) * 0.1
   if theta > pi then return nil end
   local s = sin(theta) / r
   return x * s, y * s, cos(theta)
end
