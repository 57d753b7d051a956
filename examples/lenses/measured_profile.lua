-- A lens from a calibration table: image radius -> angle from the optical axis, read while the script loads.
-- (paths are relative to the engine's working directory, as for any Lua io.open)
local radius, angle = {}, {}
local f = assert(io.open("examples/lenses/measured_profile.txt", "r"), "measured_profile.txt not found")
f:read("*l")                                   -- the header line
while true do
   local r, a = f:read("*n", "*n")
   if not r then break end
   radius[#radius + 1] = r
   angle[#angle + 1] = a
end
f:close()

max_fov = 220
max_vfov = 220
lens_width = 2
lens_height = 2
onload = "f_contain"

function lens_inverse(x, y)
   local r = sqrt(x * x + y * y)
   if r > radius[#radius] then return nil end
   if r == 0 then return 0, 0, 1 end
   local theta = 0
   for i = 1, #radius - 1 do
      if r >= radius[i] and r <= radius[i + 1] then
         local t = (r - radius[i]) / (radius[i + 1] - radius[i])
         theta = angle[i] + (angle[i + 1] - angle[i]) * t
      end
   end
   local s = sin(theta) / r
   return x * s, y * s, cos(theta)
end
